"""Pin the oracle (CPU restatement) against outputs of the REFERENCE run in the build container
(tests/golden/*.pt, produced by tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import layers as OL
from oracle import flux as OF
from tests.golden.seeded import seeded, synthetic_state_dict


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def test_sdpa_matches_reference(golden_dir):
    cases = _load(golden_dir, "attention_sdpa.pt")
    assert len(cases) == 8
    for c in cases:
        dt = torch.float32 if "float32" in c["dtype"] else torch.bfloat16
        q = seeded(c["q_shape"], c["seed"], dt)
        k = seeded(c["k_shape"], c["seed"] + 100, dt)
        v = seeded(c["k_shape"], c["seed"] + 200, dt)
        out = OL.sdpa(q, k, v)
        ref = c["out"]
        tol = 2e-6 if dt == torch.float32 else 8e-3  # bf16: one output ulp
        assert out.dtype == ref.dtype
        assert torch.allclose(out.float(), ref.float(), atol=tol, rtol=tol), c["q_shape"]


def test_efficiency_ops_match_reference(golden_dir):
    g = _load(golden_dir, "efficiency_ops.pt")
    # apply_gate_inplace: x *= gate   (efficiency/ops.py:19-34)
    c = g["gate"]
    assert torch.equal((c["x"] * c["gate"]), c["out"])
    # apply_scale_shift_inplace: x + x*scale + shift in bf16 (ops.py:37-56)
    c = g["scale_shift"]
    x = c["x"].clone()
    x = x + x * c["scale"]
    x = x + c["shift"]
    assert torch.allclose(x.float(), c["out"].float(), atol=4e-2, rtol=2e-2)
    # InplaceRMSNorm on bf16 input == intended RMSNorm semantics to bf16 rounding (App. B-2)
    c = g["rmsnorm_bf16"]
    n = OL.RMSNorm(256, eps=c["eps"])
    with torch.no_grad():
        n.weight.copy_(c["weight"])
    y = n(c["x"])
    assert torch.allclose(y.float(), c["out"].float(), atol=3e-2, rtol=2e-2)


def test_flux_wiring_matches_reference_blocks(golden_dir):
    g = _load(golden_dir, "flux_hybrid.pt")
    model = OF.FluxTransformer2DModel(**g["config"]).eval()
    assert sorted(model.state_dict().keys()) == g["keys"]
    model.load_state_dict(synthetic_state_dict(model, g["seed"]), strict=True)
    inp = g["inputs"]
    out = model(inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"],
                inp["timestep"], inp["img_ids"], inp["txt_ids"], inp["guidance"])
    ref = g["out"]
    rel = (out - ref).norm() / ref.norm()
    assert rel < 1e-5, rel
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)


def test_flux_controlnet_residual_placement_matches_reference(golden_dir):
    """`controlnet_block_samples` / `controlnet_single_block_samples` / `controlnet_blocks_repeat` of the reference Flux forward
    (transformer/flux/base/model.py:594-640) — 3 + 3 blocks, 2 + 2 samples: oracle.flux vs the reference run."""
    g = _load(golden_dir, "flux_controlnet.pt")
    model = OF.FluxTransformer2DModel(**g["config"]).eval()
    assert sorted(model.state_dict().keys()) == g["keys"]
    model.load_state_dict(synthetic_state_dict(model, g["seed"]), strict=True)
    inp = g["inputs"]
    n_img, dim = inp["hidden_states"].shape[1], model.inner_dim
    cd = [seeded((1, n_img, dim), s) * g["scale"] for s in g["double_seeds"]]
    cs = [seeded((1, n_img, dim), s) * g["scale"] for s in g["single_seeds"]]
    args = (inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"], inp["timestep"], inp["img_ids"],
            inp["txt_ids"], inp["guidance"])
    for name, kw in (("interval", dict(controlnet_block_samples=cd, controlnet_single_block_samples=cs)),
                     ("repeat", dict(controlnet_block_samples=cd, controlnet_blocks_repeat=True)), ("none", {})):
        out = model(*args, **kw)
        rel = float((out - g["out"][name]).norm() / g["out"][name].norm())
        assert rel < 1e-5, (name, rel)


def test_bf16_storage_policy_is_close_to_fp32(golden_dir):
    g = _load(golden_dir, "flux_hybrid.pt")
    model = OF.FluxTransformer2DModel(**g["config"]).eval()
    model.load_state_dict(synthetic_state_dict(model, g["seed"]), strict=True)
    inp = g["inputs"]
    args = (inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"],
            inp["timestep"], inp["img_ids"], inp["txt_ids"], inp["guidance"])
    a = model(*args)
    b = model(*args, policy=OL.BF16_STORAGE)
    rel = (a - b).norm() / a.norm()
    assert 0 < rel < 3e-2, rel


def test_pack_unpack_roundtrip():
    x = torch.arange(2 * 16 * 8 * 12, dtype=torch.float32).reshape(2, 16, 8, 12)
    p = OF.pack_latents(x)
    assert p.shape == (2, 4 * 6, 64)
    assert torch.equal(OF.unpack_latents(p, 64, 96, 8), x)
    assert abs(OF.calculate_shift(4096) - 1.15) < 1e-9 and abs(OF.calculate_shift(256) - 0.5) < 1e-9


def test_wan_wiring_matches_reference_blocks(golden_dir):
    """oracle.wan vs the reference's own WanTransformer3DModel run in float64 (intended RMSNorm)."""
    from oracle import wan as OW
    g = _load(golden_dir, "wan_hybrid.pt")
    model = OW.WanTransformer3DModel(**g["config"]).eval()
    assert sorted(model.state_dict().keys()) == g["keys"]
    model.load_state_dict(synthetic_state_dict(model, g["seed"]), strict=True)
    inp = g["inputs"]
    out = model(inp["hidden_states"], inp["timestep"], inp["encoder_hidden_states"])
    rel = (out - g["out"]).norm() / g["out"].norm()
    assert out.shape == g["out"].shape and rel < 2e-5, rel
    b = model(inp["hidden_states"], inp["timestep"], inp["encoder_hidden_states"], policy=OL.BF16_STORAGE)
    assert 0 < (out - b).norm() / out.norm() < 3e-2


def test_qwen_wiring_matches_reference_blocks(golden_dir):
    """oracle.qwenimage vs the reference's own QwenImageTransformer2DModel (edit layout, two images)."""
    from oracle import qwenimage as OQ
    g = _load(golden_dir, "qwen_hybrid.pt")
    model = OQ.QwenImageTransformer2DModel(**g["config"]).eval()
    assert sorted(model.state_dict().keys()) == g["keys"]
    model.load_state_dict(synthetic_state_dict(model, g["seed"]), strict=True)
    inp = g["inputs"]
    out = model(inp["hidden_states"], inp["encoder_hidden_states"], inp["timestep"], inp["img_shapes"])
    rel = (out - g["out"]).norm() / g["out"].norm()
    assert out.shape == g["out"].shape and rel < 2e-5, rel
    pos = OQ.qwen_rope_positions([(1, 6, 8), (1, 4, 6)], 13)
    assert pos.shape == (13 + 48 + 24, 3)
    assert pos[0].tolist() == [4, 4, 4] and pos[13].tolist() == [0, -3, -4] and pos[13 + 48].tolist() == [1, -2, -3]


def test_vae_full_sequence_restatement_matches_streaming_reference(golden_dir):
    """oracle.vae_wan decodes a tile in ONE causal pass; the reference streams frame by frame with
    feat_cache (tests/golden/vae_wan.pt is the reference's output).  They must agree: this pins the
    reformulation the HIP path uses, including the "Rep" first-frame rule and the in-place tile blends."""
    from oracle.vae_wan import AutoencoderKLWanDecoder
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_wan.pt")
    vae = AutoencoderKLWanDecoder(**g["config"]).eval()
    assert sorted(vae.state_dict().keys()) == g["keys"]
    vae.load_state_dict(vae_synthetic_state_dict(vae, g["seed"]), strict=True)
    z = seeded(g["z_shape"], g["z_seed"])
    untiled = vae.decode(z)
    assert untiled.shape == g["untiled"].shape
    assert torch.allclose(untiled[0, :, :, ::8, ::8], g["untiled_f32_sample"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(untiled, g["untiled"].float(), atol=8e-3, rtol=8e-3)     # golden stored in bf16
    vae.enable_tiling(*g["tile"])
    tiled = vae.decode(z)
    assert torch.allclose(tiled[0, :, :, ::8, ::8], g["tiled_f32_sample"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(tiled, g["tiled"].float(), atol=8e-3, rtol=8e-3)
    assert float((tiled - untiled).abs().max()) > 1e-3


def test_hunyuan15_vae_restatement_matches_reference(golden_dir):
    """oracle.vae_hunyuan15 against the reference AutoencoderKLHunyuanVideo15.decode (tests/golden/vae_hunyuan15.pt):
    replicate-padded causal convs, frame-causal mid-block attention, DCAE pixel-shuffle upsampling with the first-frame
    half-channel rule, and the 8x8-latent tiled decode with 32-px blends."""
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_hunyuan15.pt")
    vae = AutoencoderKLHunyuanVideo15(**g["config"]).eval()
    assert sorted(k for k in vae.state_dict() if k.startswith("decoder.")) == g["keys"]
    vae.load_state_dict(vae_synthetic_state_dict(vae, g["seed"]), strict=True)
    z = seeded(g["z_shape"], g["z_seed"])
    untiled = vae.decode(z)
    assert untiled.shape == g["untiled"].shape
    assert torch.allclose(untiled[0, :, :, ::8, ::8], g["untiled_f32_sample"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(untiled, g["untiled"].float(), atol=8e-3, rtol=8e-3)     # golden stored in bf16
    vae.enable_tiling()
    tiled = vae.decode(z)
    assert torch.allclose(tiled[0, :, :, ::8, ::8], g["tiled_f32_sample"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(tiled, g["tiled"].float(), atol=8e-3, rtol=8e-3)
    assert float((tiled - untiled).abs().max()) > 1e-3


def test_hunyuan15_vae_encoder_restatement_matches_reference(golden_dir):
    """oracle.vae_hunyuan15 ENCODE against the reference AutoencoderKLHunyuanVideo15._encode (vae_hunyuan15_encode.pt): DCAE
    pixel un-shuffle downsampling with the grouped-mean shortcut, the first-frame rule of the temporal downsamplers, the
    grouped-mean shortcut around conv_out, and the tiled encode with latent-space blends."""
    from oracle.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_hunyuan15_encode.pt")
    vae = AutoencoderKLHunyuanVideo15(**g["config"]).eval()
    assert sorted(vae.state_dict().keys()) == g["keys"]
    vae.load_state_dict(vae_synthetic_state_dict(vae, g["seed"]), strict=True)
    for name in ("image", "clip"):
        c = g[name]
        out = vae.encode(seeded(c["shape"], c["seed"]).clamp(-1, 1))
        assert out.shape == c["moments"].shape
        assert torch.allclose(out, c["moments"], atol=2e-5, rtol=1e-4), (name, float((out - c["moments"]).abs().max()))
    c = g["tiled"]
    vae.enable_tiling()
    out = vae.encode(seeded(c["shape"], c["seed"]).clamp(-1, 1), tile_sample_min=c["tile"])
    assert torch.allclose(out, c["moments"], atol=2e-5, rtol=1e-4), float((out - c["moments"]).abs().max())
    untiled = AutoencoderKLHunyuanVideo15.encode(vae, seeded(c["shape"], c["seed"]).clamp(-1, 1), tile_sample_min=4096)
    assert float((untiled - out).abs().max()) > 1e-3, "tiling must be observable"


def test_taehv_light_vae_restatement_matches_reference(golden_dir):
    """oracle.vae_taehv against the reference's TAEHV decoder wrapped by AutoencoderKLHunyuanVideo15Light (vae_taehv.pt, run
    in sequential mode as the engine does; the parallel mode agreed to the recorded f32 noise): MemBlock memory = the
    previous frame (zeros first), TGrow's channel blocks -> frames, nearest 2x upsamples, leaky ReLU 0.2, clamp, pixel
    shuffle, 3 trimmed frames, and the `taehv.decoder.*` key set."""
    from oracle.vae_taehv import AutoencoderKLHunyuanVideo15Light
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_taehv.pt")
    vae = AutoencoderKLHunyuanVideo15Light(scaling_factor=g["scaling_factor"]).eval()
    assert sorted(vae.state_dict().keys()) == g["keys"]
    assert set(g["keys"]) < set(g["all_keys"]) and all(k.startswith("taehv.encoder.") for k in set(g["all_keys"]) - set(g["keys"]))
    vae.load_state_dict(vae_synthetic_state_dict(vae, g["seed"]), strict=True)
    for name in ("clip", "frame"):
        c = g[name]
        out = vae.decode(seeded(c["shape"], c["seed"]) * c["scale"])
        ref = c["sequential"]
        assert out.shape == ref.shape and out.shape[2] == 4 * c["shape"][2] - 3
        assert c["parallel_max_abs_diff"] < 1e-5
        assert torch.allclose(out, ref, atol=2e-5, rtol=1e-4), (name, float((out - ref).abs().max()))
        assert float(ref.abs().max()) <= 1.0 and float(ref.std()) > 0.05, "the fixture must exercise the decoder, not the clamp"


def test_hunyuan15_wiring_matches_reference_blocks(golden_dir):
    """oracle.hunyuan15 against the reference's own HunyuanVideo-1.5 classes (hybrid oracle, float64 run): token
    refiner with a key-padding mask, t2v and i2v token orders, RoPE on latent tokens only, un-patchify."""
    from oracle.hunyuan15 import HunyuanVideo15Transformer3DModel
    g = torch.load(os.path.join(golden_dir, "hunyuan15_hybrid.pt"), weights_only=False)
    m = HunyuanVideo15Transformer3DModel(**g["config"]).eval()
    sd = synthetic_state_dict(m, g["seed"])
    assert sorted(sd.keys()) == g["keys"]
    m.load_state_dict(sd)
    i = g["inputs"]
    for name, img in (("t2v", torch.zeros_like(g["image_embeds_i2v"])), ("i2v", g["image_embeds_i2v"])):
        out = m(i["hidden_states"], i["timestep"], i["encoder_hidden_states"], i["encoder_attention_mask"],
                i["encoder_hidden_states_2"], i["encoder_attention_mask_2"], img)
        ref = g["out"][name]
        rel = float((out - ref).norm() / ref.norm())
        assert rel < 1e-5, (name, rel)


def _text_sd(model, seed, norm_seed, tag):
    from tests.golden.seeded import text_encoder_state_dict
    sd = text_encoder_state_dict(model, seed, norm_seed, tag)
    sd.pop("encoder.embed_tokens.weight", None)
    return sd


def test_text_encoders_restatement_matches_transformers(golden_dir):
    """oracle.text_encoders against the `transformers` classes the reference loads by name (text_encoder.py:24-82):
    T5 (gated-gelu and relu feed-forward), UMT5 (per-layer position bias), CLIP text (causal, quick-gelu, EOS pooling),
    with and without a padding mask (tests/golden/text_encoders.pt, generated from the package installed here)."""
    from oracle import text_encoders as OT
    g = torch.load(os.path.join(golden_dir, "text_encoders.pt"), weights_only=False)
    ids, mask = g["t5_ids"], g["t5_mask"]
    for name, extra in (("t5", {}), ("t5_relu", dict(feed_forward_proj="relu")), ("umt5", dict(per_layer_bias=True))):
        c = g[name]
        m = OT.T5EncoderModel(**{**g["t5_config"], **extra}).eval()
        assert sorted(m.state_dict().keys()) == sorted(c["keys"] + ["encoder.embed_tokens.weight"]), name
        sd = _text_sd(m, c["seed"], 24, "layer_norm.weight")
        m.load_state_dict(sd, strict=False)
        out = m(ids)
        assert len(out.hidden_states) == c["n_hidden"]
        assert torch.allclose(out.last_hidden_state, c["last"], atol=2e-5, rtol=1e-4), name
        assert torch.allclose(out.hidden_states[1], c["hidden1"], atol=2e-5, rtol=1e-4), name
        outm = m(ids, attention_mask=mask)
        real = mask.bool()
        assert torch.allclose(outm.last_hidden_state[real], c["last_masked"][real], atol=2e-5, rtol=1e-4), name
    c = g["clip"]
    m = OT.CLIPTextModel(**g["clip_config"]).eval()
    assert sorted(m.state_dict().keys()) == c["keys"]
    m.load_state_dict(_text_sd(m, c["seed"], 30, "layer_norm"), strict=True)
    out = m(g["clip_ids"])
    assert len(out.hidden_states) == c["n_hidden"]
    assert torch.allclose(out.last_hidden_state, c["last"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(out.pooler_output, c["pooled"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(out.hidden_states[-2], c["hidden_m2"], atol=2e-5, rtol=1e-4)
    outm = m(g["clip_ids"], attention_mask=g["clip_mask"])
    real = g["clip_mask"].bool()
    assert torch.allclose(outm.last_hidden_state[real], c["last_masked"][real], atol=2e-5, rtol=1e-4)
    assert torch.allclose(outm.pooler_output, c["pooled_masked"], atol=2e-5, rtol=1e-4)


def _qwen_vl_oracle(g):
    from oracle.qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from tests.golden.seeded import text_encoder_state_dict
    m = Qwen2_5_VLForConditionalGeneration(**g["text_config"], mrope_section=(16, 24, 24), image_token_id=g["image_token_id"],
                                           vision_config=g["vision_config"]).eval()
    sd = text_encoder_state_dict(m, g["seed"], 52, "norm")
    for k in [k for k in sd if k.endswith("ln_q.weight")]:
        sd[k] = 1.0 + 0.1 * seeded(sd[k].shape, 53).to(torch.bfloat16).float()
    assert sorted(sd.keys()) == g["keys"]
    m.load_state_dict(sd, strict=True)
    return m, sd


def test_qwen2_5_vl_restatement_matches_transformers(golden_dir):
    """oracle.qwen2_5_vl against transformers.Qwen2_5_VLForConditionalGeneration (tests/golden/qwen2_5_vl.pt): the causal
    GQA decoder with a right-padded batch, and a two-image prompt through the vision tower (window index, 2-D rotary,
    windowed / full blocks, merger), the embedding scatter and the 3-D RoPE position ids."""
    from oracle import qwen2_5_vl as OQV
    g = torch.load(os.path.join(golden_dir, "qwen2_5_vl.pt"), weights_only=False)
    m, _ = _qwen_vl_oracle(g)
    t = g["text"]
    out = m(t["ids"], attention_mask=t["mask"])
    real = t["mask"].bool()
    assert len(out.hidden_states) == t["n_hidden"]
    assert torch.allclose(out.hidden_states[-1][real], t["last"][real], atol=3e-5, rtol=1e-4)
    assert torch.allclose(out.hidden_states[1][real], t["hidden1"][real], atol=3e-5, rtol=1e-4)
    im = g["image"]
    pos = OQV.rope_index(im["ids"], im["mask"], im["grid"], g["image_token_id"], g["vision_config"]["spatial_merge_size"])
    assert torch.equal(pos, im["position_ids"])
    vis = m.model.visual(im["pixel_values"], im["grid"])
    assert torch.allclose(vis, im["vision"], atol=3e-5, rtol=1e-4)
    out = m(im["ids"], attention_mask=im["mask"], pixel_values=im["pixel_values"], image_grid_thw=im["grid"])
    assert torch.allclose(out.hidden_states[-1], im["last"], atol=3e-5, rtol=1e-4)


def test_vae_encode_full_sequence_restatement_matches_streaming_reference(golden_dir):
    """oracle.vae_wan.AutoencoderKLWanEncoder encodes a clip in ONE pass (strided temporal downsampling over the whole
    sequence, frame 0 passing through); the reference streams 1 + 4 + 4 frames with feat_cache
    (tests/golden/vae_wan_encode.pt is its `_encode` output: clip and single image, untiled and tiled)."""
    from oracle.vae_wan import AutoencoderKLWanEncoder
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_wan_encode.pt")
    vae = AutoencoderKLWanEncoder(**g["config"]).eval()
    assert sorted(vae.state_dict().keys()) == g["keys"]
    vae.load_state_dict(vae_synthetic_state_dict(vae, g["seed"]), strict=True)
    x = seeded(g["x_shape"], g["x_seed"])
    assert torch.allclose(vae.encode(x), g["video"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(vae.encode(x[:, :, :1]), g["image"], atol=2e-5, rtol=1e-4)
    vae.enable_tiling(*g["tile"])
    assert torch.allclose(vae.encode(x), g["video_tiled"], atol=2e-5, rtol=1e-4)
    assert torch.allclose(vae.encode(x[:, :, :1]), g["image_tiled"], atol=2e-5, rtol=1e-4)


def test_hunyuan15_meanflow_time_embedding_matches_reference(golden_dir):
    """oracle.hunyuan15 built with use_meanflow=True against the reference class (hunyuan15_meanflow.pt, float64 run): the
    second timestep embedder's output is added to temb when `timestep_r` is given and ignored when it is None."""
    from oracle.hunyuan15 import HunyuanVideo15Transformer3DModel
    g = torch.load(os.path.join(golden_dir, "hunyuan15_meanflow.pt"), weights_only=False)
    m = HunyuanVideo15Transformer3DModel(**g["config"]).eval()
    sd = synthetic_state_dict(m, g["seed"])
    assert sorted(sd.keys()) == g["keys"] and any("timestep_embedder_r" in k for k in g["keys"])
    m.load_state_dict(sd)
    i = g["inputs"]
    for name, tr in (("r300", g["timestep_r"]), ("none", None)):
        out = m(i["hidden_states"], i["timestep"], i["encoder_hidden_states"], i["encoder_attention_mask"],
                i["encoder_hidden_states_2"], i["encoder_attention_mask_2"], g["image_embeds"], timestep_r=tr)
        ref = g["out"][name]
        rel = float((out - ref).norm() / ref.norm())
        assert rel < 1e-5, (name, rel)
    assert float((g["out"]["r300"] - g["out"]["none"]).norm() / g["out"]["none"].norm()) > 1e-2


def test_taehv_encoder_restatement_matches_reference(golden_dir):
    """oracle.vae_taehv.TAEHVEncoder against the reference TAEHV.encode_video (vae_taehv_encode.pt; its sequential graph walk
    agreed with the parallel mode to 1e-6): pixel un-shuffle, last-frame padding to a multiple of 4, TPool = frame pairs stacked
    along the channels + 1x1 conv, stride-2 convs, MemBlocks."""
    from oracle.vae_taehv import TAEHVEncoder
    from tests.golden.seeded import vae_synthetic_state_dict
    g = _load(golden_dir, "vae_taehv_encode.pt")
    enc = TAEHVEncoder().eval()
    assert sorted(enc.state_dict().keys()) == g["keys"]
    enc.load_state_dict(vae_synthetic_state_dict(enc, g["seed"]), strict=True)
    for name in ("clip9", "clip4"):
        c = g[name]
        with torch.no_grad():
            out = enc.encode_video((seeded(c["shape"], c["seed"]) * 0.25 + 0.5).clamp(0, 1))
        assert out.shape == c["latents"].shape and c["sequential_max_abs_diff"] < 1e-5
        assert torch.allclose(out, c["latents"], atol=2e-5, rtol=1e-4), (name, float((out - c["latents"]).abs().max()))


@pytest.mark.skipif(not os.path.isdir("/root/reference/apps/api"), reason="the reference is only present in the build container")
def test_fixture_recipe_reproduces_the_committed_files():
    """`make_golden.py --check all`: every generator is re-run in ONE fresh process against /root/reference (in the order that
    used to break `gen_convert`) and every file written must equal the committed fixture bit for bit (≈20 s)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py"), "--check", "all"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "26 generators -> 26 files compared, 0 mismatches" in r.stdout, tail


def test_easycache_restatement_matches_the_reference_function(golden_dir):
    """oracle.easycache (the step-skipping rule of the reference's `easycache_forward_`, transformer/wan/base/model.py:202-520)
    around the oracle Wan model (fp32) against the reference function itself run on the reference model
    (float64, tests/golden/wan_easycache.pt): the same calls are computed / served from the cache, every output to 5e-5."""
    from oracle import wan as OW
    from oracle.easycache import EasyCacheState, easycache_forward
    g = _load(golden_dir, "wan_easycache.pt")
    m = OW.WanTransformer3DModel(**g["config"]).eval()
    m.load_state_dict(synthetic_state_dict(m, g["seed"]), strict=True)
    x = seeded((1, 16, 3, 8, 12), g["x_seed"])
    txts = [seeded((1, 20, 64), s) for s in g["txt_seeds"]]
    st = EasyCacheState(g["n"], g["thresh"], g["ret_steps"])
    computed, k = [], 0
    with torch.no_grad():
        for i in range(g["n"]):
            pair = []
            for txt in txts:
                t = torch.tensor([g["timesteps"][i]])
                out, did = easycache_forward(st, lambda: m(x, t, txt), x, g["config"]["out_channels"])
                computed.append(did)
                ref = g["outs"][k]
                rel = float((out - ref).norm() / ref.norm())
                assert out.dtype == torch.float32 and rel < 5e-5, (k, rel)
                pair.append(out)
                k += 1
            x = x - g["dt"] * (pair[1] + g["guidance"] * (pair[0] - pair[1]))
    assert computed == g["computed"] and not all(computed), "".join("C" if c else "-" for c in computed)
    assert float((x.float() - g["x_final"]).norm() / g["x_final"].norm()) < 5e-5


@pytest.mark.parametrize("case", ["zero_cond_t", "additional_t_cond", "both"])
def test_oracle_qwen_variants_match_the_reference_run(golden_dir, case):
    """oracle.qwenimage with `zero_cond_t` / `use_additional_t_cond` against the reference class run here (qwen_variants.pt)."""
    from oracle.qwenimage import QwenImageTransformer2DModel
    g = torch.load(os.path.join(golden_dir, "qwen_variants.pt"), weights_only=False)
    c, inp = g["cases"][case], g["inputs"]
    m = QwenImageTransformer2DModel(**c["config"]).eval()
    sd = synthetic_state_dict(m, g["seed"])
    assert sorted(sd.keys()) == c["keys"]
    m.load_state_dict(sd, strict=True)
    out = m(inp["hidden_states"], inp["encoder_hidden_states"], inp["timestep"], inp["img_shapes"], additional_t_cond=c["additional_t_cond"])
    rel = float((out - c["out"]).norm() / c["out"].norm())
    assert rel < 1e-5, rel


@pytest.mark.parametrize("case", ["one", "two"])
def test_oracle_flux_ip_adapter_matches_the_reference_run(golden_dir, case):
    """oracle.flux with IP-adapter processors on the double blocks against the reference model run with its
    FluxIPAdapterAttnProcessor (flux_ip_adapter.pt)."""
    from oracle.flux import FluxTransformer2DModel, FluxIPAdapterProcessor
    g = torch.load(os.path.join(golden_dir, "flux_ip_adapter.pt"), weights_only=False)
    cfg, inp, c = g["config"], g["inputs"], g["cases"][case]
    dim = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    m = FluxTransformer2DModel(**cfg).eval()
    for blk in m.transformer_blocks:
        blk.attn.processor = FluxIPAdapterProcessor(dim, cfg["joint_attention_dim"], c["num_tokens"], c["scale"])
    sd = synthetic_state_dict(m, g["seed"])
    assert sorted(sd.keys()) == c["keys"]
    m.load_state_dict(sd, strict=True)
    ips = [seeded((1, n, cfg["joint_attention_dim"]), s) for n, s in zip(c["num_tokens"], c["ip_seeds"])]
    out = m(inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"], inp["timestep"], inp["img_ids"],
            inp["txt_ids"], inp["guidance"], ip_hidden_states=ips)
    rel = float((out - c["out"]).norm() / c["out"].norm())
    assert rel < 1e-5, rel
