"""LoRA: key normalisation / alpha folding against the reference's own LoraConverter output
(tests/golden/lora_convert.pt), the oracle's merged form against its runtime form, host-side adapter bookkeeping,
and — on the GPU — the one-GEMM merge and a LoRA'd Flux forward against the oracle."""
import os

import pytest
import torch

from oracle import lora as OLR
from tests.golden.seeded import seeded

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lora_convert.pt")


def test_normalize_matches_reference_converter():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora
    g = torch.load(GOLD, weights_only=False)
    for name, case in g["cases"].items():
        got = lora.normalize_lora_state_dict({k: v.clone() for k, v in case["inp"].items()})
        ref = case["out"]
        assert set(got) == set(ref), (name, sorted(set(got) ^ set(ref)))
        for k in ref:
            assert torch.equal(got[k], ref[k]), (name, k)      # bit-exact: same multiplications in the same order
    for (r, a), (sd, su) in g["alpha_scales"].items():
        assert lora.alpha_scales(r, a) == (sd, su) == OLR.get_alpha_scales(r, a)


def test_oracle_fold_alpha_matches_reference():
    g = torch.load(GOLD, weights_only=False)
    case = g["cases"]["peft"]
    got = OLR.fold_alpha({k: v.clone() for k, v in case["inp"].items()})
    for k, v in case["out"].items():
        assert torch.equal(got[k], v), k


def test_unsupported_formats_and_shapes_raise():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora
    # Kohya single-file keys are accepted since round 3 (restatement of lora_converter.py:185-255, tests/test_converters.py)
    out = lora.normalize_lora_state_dict({"lora_unet_double_blocks_0_img_attn_proj.lora_down.weight": torch.zeros(4, 8)})
    assert list(out) == ["unet.double_blocks.0.img_attn.proj.lora_A.weight"]
    with pytest.raises(ValueError, match="diffusers_old"):
        lora.normalize_lora_state_dict({"blocks.0.attn.to_q_lora.down.weight": torch.zeros(4, 8)})
    with pytest.raises(ValueError, match="lacks"):
        lora.split_modules({"a.lora_A.weight": torch.zeros(4, 8)})
    with pytest.raises(ValueError, match="DoRA"):
        lora.split_modules({"a.lora_magnitude_vector": torch.zeros(8)})
    mods = lora.split_modules({"transformer.a.b.lora_A.weight": torch.zeros(4, 8), "transformer.a.b.lora_B.weight": torch.zeros(16, 4)})
    assert list(mods) == ["a.b"]                                  # unanimous known prefix is stripped


def test_oracle_merged_equals_runtime_form():
    x = seeded((5, 32), 1)
    w, b = seeded((48, 32), 2), seeded((48,), 3)
    ads = [(seeded((8, 32), 4), seeded((48, 8), 5), 0.7), (seeded((4, 32), 6), seeded((48, 4), 7), -1.3)]
    y_rt = OLR.lora_linear(x, w, b, ads)
    y_mg = torch.nn.functional.linear(x, OLR.merged_weight(w, ads), b)
    assert torch.allclose(y_rt, y_mg, rtol=1e-5, atol=1e-5)


class _Host(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(32, 48)


def test_adapter_bookkeeping_needs_gpu_and_validates():
    """No CPU fallback: merging on a CPU model raises; wrong targets / shapes raise before anything is touched."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora

    class M(lora.LoraAdapterMixin, _Host):
        pass
    m = M()
    sd = {"a.lora_A.weight": seeded((4, 32), 1), "a.lora_B.weight": seeded((48, 4), 2)}
    with pytest.raises(KeyError):
        m.load_lora_adapter({"zz.lora_A.weight": sd["a.lora_A.weight"], "zz.lora_B.weight": sd["a.lora_B.weight"]})
    with pytest.raises(ValueError, match="does not fit"):
        m.load_lora_adapter({"a.lora_A.weight": seeded((4, 16), 1), "a.lora_B.weight": seeded((48, 4), 2)})
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            m.load_lora_adapter(sd, adapter_name="x")
    m2 = M()
    m2.load_lora_adapter(sd, adapter_name="x", activate=False)      # deferred: nothing merged yet
    assert m2.active_adapters() == []
    with pytest.raises(ValueError, match="already loaded"):
        m2.load_lora_adapter(sd, adapter_name="x", activate=False)
    with pytest.raises(ValueError, match="not loaded"):
        m2.set_adapters(["nope"])


# ---------------------------------------------------------------- GPU
DEV = "cuda"


@pytest.mark.gpu
def test_merge_weight_one_gemm_matches_oracle():
    """W = base + sum s_i B_i A_i on the HIP GEMM (row-range view of a packed matrix, two adapters, ragged rank).
    Bar: equal to the oracle fed the same bf16 factors up to the final bf16 rounding (<= 1 ulp of the result)."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora
    packed = seeded((3 * 384, 256), 11).mul(0.05).to(torch.bfloat16).to(DEV)
    view = packed[384:768]                                           # the "to_k" rows of a fused QKV weight
    base = view.clone()
    ads = [(seeded((8, 256), 12), seeded((384, 8), 13).mul(0.1), 0.8),
           (seeded((20, 256), 14), seeded((384, 20), 15).mul(0.1), -0.5)]
    before = packed.clone()
    lora.merge_weight_(view, base, ads)
    torch.cuda.synchronize()
    ref = OLR.merged_weight(base.float().cpu(), ads, factors_bf16=True)
    got = view.float().cpu()
    ulp = ref.abs().clamp_min(2.0 ** -126) * 2.0 ** -8
    assert ((got - ref).abs() <= ulp).all(), float((got - ref).abs().max())
    assert torch.equal(packed[:384], before[:384]) and torch.equal(packed[768:], before[768:])   # neighbours untouched
    # and against the un-rounded factors: the bf16 factor rounding stays inside 2e-3 of the delta's size
    exact = OLR.merged_weight(base.float().cpu(), ads)
    delta = (exact - base.float().cpu())
    assert float((got - exact).norm() / delta.norm()) < 2e-2


@pytest.mark.gpu
def test_flux_with_lora_matches_oracle_and_restores():
    """Tiny Flux, adapters on a fused-QKV member, an MLP weight and an AdaLN projection; set_adapters weights,
    disable/enable and delete restore the base weights bit-exactly."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora
    from apex_studio_amd.flux import FluxTransformer2DModel
    from oracle import flux as OF
    from oracle import layers as OL
    from tests.golden.seeded import synthetic_state_dict
    cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128,
               num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True,
               axes_dims_rope=(16, 56, 56))
    orc = OF.FluxTransformer2DModel(**cfg).eval()
    sd = synthetic_state_dict(orc, 7)
    targets = ["transformer_blocks.0.attn.to_k", "transformer_blocks.1.ff.net.0.proj",
               "single_transformer_blocks.0.proj_out", "transformer_blocks.0.norm1.linear"]
    lsd, ads = {}, {}
    for i, t in enumerate(targets):
        w = sd[t + ".weight"]
        a, b = seeded((8, w.shape[1]), 300 + i).mul(0.05), seeded((w.shape[0], 8), 400 + i).mul(0.05)
        lsd[f"transformer.{t}.lora_A.weight"], lsd[f"transformer.{t}.lora_B.weight"] = a, b
        ads[t] = (a, b)
    m = FluxTransformer2DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    h2 = w2 = 8
    inp = dict(hidden_states=seeded((1, 64, 64), 31), encoder_hidden_states=seeded((1, 16, 128), 32),
               pooled_projections=seeded((1, 64), 33), timestep=torch.tensor([0.5]), guidance=torch.tensor([4.0]),
               img_ids=OF.latent_image_ids(h2, w2), txt_ids=torch.zeros(16, 3))
    g = {k: (v.to(DEV).to(torch.bfloat16) if k in ("hidden_states", "encoder_hidden_states", "pooled_projections")
             else v.to(DEV)) for k, v in inp.items()}
    y0 = m(return_dict=False, **g)[0].float().cpu()
    base_w = {t: dict(m.named_parameters())[t + ".weight"].detach().clone() for t in targets}

    names = lora.apply_loras(m, [(lsd, 0.6)], adapter_names=["style"])
    assert names == ["style"] and m.active_adapters() == ["style"]
    y1 = m(return_dict=False, **g)[0].float().cpu()
    # oracle with merged weights (bf16-rounded weights, bf16 storage policy like the other model tests)
    sd_m = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    for t, (a, b) in ads.items():
        sd_m[t + ".weight"] = OLR.merged_weight(sd_m[t + ".weight"], [(a, b, 0.6)], factors_bf16=True).to(torch.bfloat16).float()
    orc.load_state_dict(sd_m)
    rin = {k: (v.to(torch.bfloat16).float() if k in ("hidden_states", "encoder_hidden_states", "pooled_projections")
               else v) for k, v in inp.items()}
    with torch.no_grad():
        ref = orc(rin["hidden_states"], rin["encoder_hidden_states"], rin["pooled_projections"], rin["timestep"],
                  rin["img_ids"], rin["txt_ids"], rin["guidance"], policy=OL.BF16_STORAGE).float()
    rel = float((y1 - ref).norm() / ref.norm())
    moved = float((y1 - y0).norm() / y0.norm())
    print(f"[flux lora] hip vs oracle(merged) {rel:.3e}; adapter moved the output by {moved:.3e}")
    assert rel < 1e-2 and moved > 5 * rel               # the adapter's effect is well above the parity noise

    m.set_adapters(["style"], weights=[0.0])
    for t in targets:
        assert torch.equal(dict(m.named_parameters())[t + ".weight"], base_w[t]), t
    m.set_adapters(["style"], weights=[0.6])
    m.disable_lora()
    assert torch.equal(m(return_dict=False, **g)[0].float().cpu(), y0)
    m.enable_lora()
    assert torch.equal(m(return_dict=False, **g)[0].float().cpu(), y1)      # deterministic re-merge
    m.unload_lora_weights()
    assert m.active_adapters() == [] and not m._lora_base
    assert torch.equal(m(return_dict=False, **g)[0].float().cpu(), y0)
