"""Qwen2.5-VL prompt encoder on HIP: grouped-query / block-diagonal attention and rotate-half RoPE ops
against PyTorch fp32, the vision tower and the decoder against the oracle and the `transformers` outputs in
tests/golden/qwen2_5_vl.pt."""
import os

import pytest
import torch

from oracle import layers as OL
from oracle import qwen2_5_vl as OQV
from tests.golden.seeded import seeded, text_encoder_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("H,Hkv,S,D,mode", [(28, 4, 150, 128, "causal_keep"), (4, 1, 77, 128, "causal"),
                                            (16, 16, 92, 128, "seg"), (6, 2, 200, 64, "seg")])
def test_attention_gqa_and_segments(H, Hkv, S, D, mode):
    from apex_studio_amd import ops
    q = _bf(seeded((S, H * D), 1)).to(DEV)
    kv = _bf(seeded((S, 2 * Hkv * D), 2)).to(DEV)
    k, v = kv[:, :Hkv * D], kv[:, Hkv * D:]
    keep = seg = None
    allowed = torch.ones(S, S, dtype=torch.bool, device=DEV)
    if "causal" in mode:
        allowed = allowed.tril()
    if "keep" in mode:
        keep = torch.ones(S, dtype=torch.uint8, device=DEV)
        keep[S - 31:] = 0
        allowed = allowed & keep.bool()[None, :]
    if mode == "seg":
        cuts = torch.tensor([0, 16, 17, 49, 64, S])
        seg = torch.bucketize(torch.arange(S), cuts[1:], right=True).to(torch.int32).to(DEV)
        allowed = allowed & (seg[:, None] == seg[None, :])
    out = ops.attention_bias(q, k, v, H, D ** -0.5, keep=keep, seg=seg, causal="causal" in mode, kv_heads=Hkv)
    qh = q.float().view(S, H, D).transpose(0, 1)
    kh = k.float().view(S, Hkv, D).transpose(0, 1).repeat_interleave(H // Hkv, dim=0)
    vh = v.float().view(S, Hkv, D).transpose(0, 1).repeat_interleave(H // Hkv, dim=0)
    sc = (qh @ kh.transpose(1, 2)) * D ** -0.5
    ref = (torch.softmax(sc.masked_fill(~allowed, float("-inf")), dim=-1) @ vh).transpose(0, 1).reshape(S, H * D)
    assert torch.isfinite(out).all() and _rel(out, ref) < 1e-2, _rel(out, ref)


def test_rope_half():
    from apex_studio_amd import ops
    S, H, slot, D = 37, 6, 128, 80
    x = _bf(seeded((S, H * slot + 64), 3)).to(DEV)
    ang = seeded((S, D // 2), 4) * 3
    emb = torch.cat((ang, ang), dim=-1)
    cos, sin = emb.cos().to(DEV), emb.sin().to(DEV)
    ref = x.float().clone()
    heads = ref[:, :H * slot].view(S, H, slot)
    rot = heads[..., :D]
    heads[..., :D] = rot * cos[:, None] + OQV.rotate_half(rot) * sin[:, None]
    ops.rope_half_(x[:, :H * slot], H, slot, cos.contiguous(), sin.contiguous())
    assert torch.equal(x[:, H * slot:].cpu(), _bf(seeded((S, H * slot + 64), 3))[:, H * slot:])        # columns past the heads
    got = x[:, :H * slot].float().view(S, H, slot)
    assert torch.equal(got[..., D:].cpu(), _bf(seeded((S, H * slot + 64), 3))[:, :H * slot].float().view(S, H, slot)[..., D:])
    assert _rel(x, ref) < 3e-3


def _models(g):
    from apex_studio_amd.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration as Hip
    orc = OQV.Qwen2_5_VLForConditionalGeneration(**g["text_config"], mrope_section=(16, 24, 24),
                                                 image_token_id=g["image_token_id"], vision_config=g["vision_config"]).eval()
    sd = text_encoder_state_dict(orc, g["seed"], 52, "norm")
    for k in [k for k in sd if k.endswith("ln_q.weight")]:
        sd[k] = 1.0 + 0.1 * seeded(sd[k].shape, 53).to(torch.bfloat16).float()
    orc.load_state_dict(sd, strict=True)
    cfg = dict(text_config={**g["text_config"], "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}},
               vision_config=g["vision_config"], image_token_id=g["image_token_id"])
    hip = Hip(cfg, device=DEV, dtype=torch.bfloat16)
    hip.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    assert sorted(hip.state_dict().keys()) == g["keys"]
    return orc, hip


def test_qwen2_5_vl_matches_transformers_and_oracle(golden_dir):
    from apex_studio_amd import qwen2_5_vl as HQ
    g = torch.load(os.path.join(golden_dir, "qwen2_5_vl.pt"), weights_only=False)
    orc, hip = _models(g)
    # text-only, right-padded batch
    t = g["text"]
    out = hip(input_ids=t["ids"].to(DEV), attention_mask=t["mask"].to(DEV), output_hidden_states=True)
    real = t["mask"].bool()
    got = out.hidden_states[-1].float().cpu()
    ref16 = orc(t["ids"], attention_mask=t["mask"], policy=OL.BF16_STORAGE).hidden_states[-1]
    ref32 = orc(t["ids"], attention_mask=t["mask"]).hidden_states[-1]
    e_like, e_ref, e_emul = _rel(got[real], ref16[real]), _rel(got[real], t["last"][real]), _rel(ref16[real], ref32[real])
    print(f"[qwen2.5-vl text] hip vs bf16-storage oracle {e_like:.3e}; vs transformers fp32 {e_ref:.3e}; emulation vs fp32 {e_emul:.3e}")
    assert len(out.hidden_states) == t["n_hidden"] and e_like < 2e-2 and e_ref < 2 * e_emul + 2e-3
    # index arithmetic: identical to transformers
    im = g["image"]
    grid = im["grid"].tolist()
    assert torch.equal(HQ.rope_index(im["ids"], im["mask"], grid, g["image_token_id"], 2), im["position_ids"])
    # vision tower alone
    vis = hip.get_image_features(im["pixel_values"].to(DEV), im["grid"]).float().cpu()
    v16 = orc.model.visual(im["pixel_values"], im["grid"], OL.BF16_STORAGE)
    v32 = orc.model.visual(im["pixel_values"], im["grid"])
    e_like, e_ref, e_emul = _rel(vis, v16), _rel(vis, im["vision"]), _rel(v16, v32)
    print(f"[qwen2.5-vl vision] hip vs bf16-storage oracle {e_like:.3e}; vs transformers fp32 {e_ref:.3e}; emulation vs fp32 {e_emul:.3e}")
    assert vis.shape == im["vision"].shape and e_like < 2e-2 and e_ref < 2 * e_emul + 2e-3
    # prompt with two images
    out = hip(input_ids=im["ids"].to(DEV), attention_mask=im["mask"].to(DEV), pixel_values=im["pixel_values"].to(DEV),
              image_grid_thw=im["grid"], output_hidden_states=True)
    got = out.hidden_states[-1].float().cpu()
    kw = dict(attention_mask=im["mask"], pixel_values=im["pixel_values"], image_grid_thw=im["grid"])
    ref16 = orc(im["ids"], policy=OL.BF16_STORAGE, **kw).hidden_states[-1]
    ref32 = orc(im["ids"], **kw).hidden_states[-1]
    e_like, e_ref, e_emul = _rel(got, ref16), _rel(got, im["last"]), _rel(ref16, ref32)
    print(f"[qwen2.5-vl text+2 images] hip vs bf16-storage oracle {e_like:.3e}; vs transformers fp32 {e_ref:.3e}; emulation vs fp32 {e_emul:.3e}")
    assert len(out.hidden_states) == im["n_hidden"] and e_like < 2e-2 and e_ref < 2 * e_emul + 2e-3
    with pytest.raises(ValueError):
        hip(input_ids=im["ids"][:, :-20].to(DEV), pixel_values=im["pixel_values"].to(DEV), image_grid_thw=im["grid"])
    with pytest.raises(NotImplementedError):     # video inputs: refused, not silently ignored
        hip(input_ids=im["ids"].to(DEV), pixel_values_videos=im["pixel_values"].to(DEV), video_grid_thw=im["grid"])
