"""Weight formats: FP-scaled dequantisation against the reference's own outputs
(tests/golden/fp_scaled.pt) and the streaming safetensors loader into the packed device layout."""
import os

import pytest
import torch

from oracle import weights as OW
from tests.golden.seeded import seeded

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fp_scaled.pt")
DT = {"e4m3fn": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}


def _same(a, b):
    """bit-equal, NaNs in the same places"""
    a, b = a.float(), b.float()
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("fmt", ["e4m3fn", "e5m2"])
def test_oracle_dequant_matches_reference(fmt):
    c = torch.load(GOLD, weights_only=False)[fmt]
    w, wall = c["w"].view(DT[fmt]), c["w_all"].view(DT[fmt])
    assert _same(OW.dequant(w, c["s_scalar"]), c["out_scalar"])
    assert _same(OW.dequant(w, c["s_scalar"]), c["out_method"])
    assert _same(OW.dequant(w, c["s_row"]), c["out_row"])
    assert _same(OW.dequant(wall, c["s_all"]), c["out_all"])
    with pytest.raises(TypeError):
        OW.dequant(w.to(torch.bfloat16), c["s_scalar"])           # the reference raises for this combination too


def test_key_map_and_iter(tmp_path):
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import weights
    from safetensors.torch import save_file
    save_file({"model.a.weight": torch.ones(2, 2), "model.b": torch.zeros(3)}, str(tmp_path / "s1.safetensors"))
    torch.save({"c": torch.ones(1)}, str(tmp_path / "s2.pt"))
    got = {weights.remap_key(k, {"model.": ""}): ld() for k, ld in
           weights.iter_checkpoint([str(tmp_path / "s1.safetensors"), str(tmp_path / "s2.pt")])}
    assert set(got) == {"a.weight", "b", "c"} and got["a.weight"].shape == (2, 2)
    lin = torch.nn.Linear(2, 2)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            weights.load_checkpoint_into(lin, [str(tmp_path / "s1.safetensors")])


# ---------------------------------------------------------------- GPU
DEV = "cuda"


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["e4m3fn", "e5m2"])
def test_hip_dequant_bit_exact_vs_reference(fmt):
    """Every fp8 code point (subnormals, max, NaN, -0) and seeded matrices, scalar and per-row scales, ragged
    width, strided output view: bit-exact against the reference fixture (integer/bit-level work: no tolerance)."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import ops
    c = torch.load(GOLD, weights_only=False)[fmt]
    w, wall = c["w"].view(DT[fmt]).to(DEV), c["w_all"].view(DT[fmt]).to(DEV)
    assert _same(ops.dequant_fp8_scaled(w, c["s_scalar"]).cpu(), c["out_scalar"])
    assert _same(ops.dequant_fp8_scaled(w, c["s_row"]).cpu(), c["out_row"])
    assert _same(ops.dequant_fp8_scaled(wall, c["s_all"]).cpu(), c["out_all"])
    wr = (seeded((7, 37), 961) * 2).to(DT[fmt])                                      # ragged: scalar tail path
    sr = seeded((7, 1), 962).abs() + 0.1
    assert _same(ops.dequant_fp8_scaled(wr.to(DEV), sr).cpu(), OW.dequant(wr, sr))
    packed = torch.zeros(3 * 24, 40, dtype=torch.bfloat16, device=DEV)                # into a packed-matrix view
    ops.dequant_fp8_scaled(w, c["s_row"], out=packed[24:48])
    assert _same(packed[24:48].cpu(), c["out_row"]) and float(packed[:24].abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="scale_weight has"):
        ops.dequant_fp8_scaled(w, torch.ones(5))


@pytest.mark.gpu
def test_streaming_loader_into_packed_flux(tmp_path):
    """Two shards (one bf16, one fp8-scaled) streamed into a packed tiny Flux: state_dict equals the expected
    tensors bit-exactly, missing/unexpected keys are reported, and the forward equals a plain load_state_dict."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import weights
    from apex_studio_amd.flux import FluxTransformer2DModel
    from oracle import flux as OF
    from safetensors.torch import save_file
    from tests.golden.seeded import synthetic_state_dict
    cfg = dict(patch_size=1, in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128,
               num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True,
               axes_dims_rope=(16, 56, 56))
    sd = {k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(OF.FluxTransformer2DModel(**cfg), 7).items()}
    keys = sorted(sd)
    q_keys = [k for k in keys if k.endswith(".weight") and sd[k].dim() == 2 and "transformer_blocks" in k][:6]
    shard_a = {k: sd[k] for k in keys[: len(keys) // 2] if k not in q_keys}
    shard_b = {k: sd[k] for k in keys[len(keys) // 2:] if k not in q_keys}
    expect = dict(sd)
    for i, k in enumerate(q_keys):                       # quantise: per-tensor scalar scale or per-row [out, 1]
        w = sd[k].float()
        s = (w.abs().amax(dim=1, keepdim=True) / 448.0) if i % 2 else (w.abs().max() / 448.0).reshape(())
        q = (w / s).to(torch.float8_e4m3fn)
        shard_b["model." + k] = q
        shard_b["model." + k[:-len("weight")] + "scale_weight"] = s.to(torch.float32)
        expect[k] = OW.dequant(q, s)
    shard_b["model.not_a_parameter.weight"] = torch.zeros(2, 2)
    missing_key = keys[0] if keys[0] not in q_keys else keys[1]
    shard_a.pop(missing_key, None)
    shard_b.pop(missing_key, None)
    save_file({("model." + k if not k.startswith("model.") else k): v.contiguous() for k, v in shard_a.items()},
              str(tmp_path / "a.safetensors"))
    save_file({("model." + k if not k.startswith("model.") else k): v.contiguous() for k, v in shard_b.items()},
              str(tmp_path / "b.safetensors"))

    m = FluxTransformer2DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    missing, unexpected = weights.load_checkpoint_into(
        m, [str(tmp_path / "a.safetensors"), str(tmp_path / "b.safetensors")], key_map={"model.": ""})
    assert missing == [missing_key] and unexpected == ["not_a_parameter.weight"]
    got = m.state_dict()
    for k, v in expect.items():
        if k != missing_key:
            assert torch.equal(got[k].cpu(), v), k
    with pytest.raises(RuntimeError, match="missing"):
        weights.load_checkpoint_into(m, [str(tmp_path / "a.safetensors")], key_map={"model.": ""}, strict=True)
    # an fp8 weight without its scale, and a scaled non-fp8 weight, are errors
    save_file({q_keys[0]: shard_b["model." + q_keys[0]]}, str(tmp_path / "c.safetensors"))
    with pytest.raises(ValueError, match="scale_weight"):
        weights.load_checkpoint_into(m, [str(tmp_path / "c.safetensors")])
    save_file({q_keys[0]: sd[q_keys[0]], q_keys[0][:-len("weight")] + "scale_weight": torch.ones(())},
              str(tmp_path / "d.safetensors"))
    with pytest.raises(TypeError, match="not fp8"):
        weights.load_checkpoint_into(m, [str(tmp_path / "d.safetensors")])


# ---------------------------------------------------------------- original-format files (round 3)
def _original_files(tmp_path, fp8_all_block_linears=False):
    """An ORIGINAL-format Wan file whose attention projections are fp8-scaled (the Kijai layout: `blocks.N.self_attn.q.weight`
    in float8_e4m3fn + `blocks.N.self_attn.q.scale_weight` + a `scaled_fp8` marker) and a BFL-format Flux file, written as
    safetensors; returns (paths, the state dicts as real tensors, model configs)."""
    from oracle import flux as OF, wan as OWan
    from safetensors.torch import save_file
    from tests.golden.make_golden_specs import flux_original_spec, wan_original_spec
    from tests.golden.seeded import spec_tensors
    wan_cfg = dict(patch_size=(1, 2, 2), num_attention_heads=1, attention_head_dim=128, in_channels=16, out_channels=16,
                   text_dim=64, freq_dim=256, ffn_dim=256, num_layers=2, cross_attn_norm=True, eps=1e-6)
    flux_cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128,
                    num_attention_heads=1, joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True,
                    axes_dims_rope=(16, 56, 56))
    wan = {k: v.to(torch.bfloat16) for k, v in spec_tensors(wan_original_spec(dim=128, ffn=256, text_dim=64, freq=256), 5000).items()}
    pick = (lambda k: k.startswith("blocks.")) if fp8_all_block_linears else (lambda k: ".self_attn." in k)
    for k in [k for k in wan if pick(k) and k.endswith(".weight") and wan[k].dim() == 2]:
        w = wan[k].float()
        s = (w.abs().max() / 448.0).reshape(())
        wan[k] = (w / s).to(torch.float8_e4m3fn)
        wan[k[:-len("weight")] + "scale_weight"] = s
    wan["scaled_fp8"] = torch.zeros(2, dtype=torch.float8_e4m3fn)
    flux = {k: v.to(torch.bfloat16) for k, v in spec_tensors(flux_original_spec(dim=128, txt=128, pooled=64), 6000).items()}
    pw, pf = str(tmp_path / "wan_orig.safetensors"), str(tmp_path / "flux_bfl.safetensors")
    save_file({k: v.contiguous() for k, v in wan.items()}, pw)
    save_file({k: v.contiguous() for k, v in flux.items()}, pf)
    return (pw, pf), (wan, flux), (wan_cfg, flux_cfg), (OWan, OF)


def test_original_format_files_stream_through_the_converter(tmp_path):
    """iter_checkpoint(converter=...): the per-file plan built on placeholders yields exactly what converting the real
    tensors yields (which tests/test_converters.py pins to the reference), fused tensors are read by row range."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import converters as CV, weights
    (pw, pf), (wan, flux), (wan_cfg, flux_cfg), (OWan, OF) = _original_files(tmp_path)
    for path, sd, conv, orc in ((pw, wan, CV.WanKeyConverter, OWan.WanTransformer3DModel(**wan_cfg)),
                                (pf, flux, CV.FluxKeyConverter, OF.FluxTransformer2DModel(**flux_cfg))):
        mk = list(orc.state_dict().keys())
        want = conv().convert(dict(sd), list(mk))
        got = {k: ld() for k, ld in weights.iter_checkpoint([path], conv(), None, mk)}
        assert sorted(got) == sorted(want)
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k].reshape(-1).view(torch.uint8), want[k].contiguous().reshape(-1).view(torch.uint8)), k
        extra = {k for k in got if k not in mk}
        assert extra == {k for k in got if k.endswith("scale_weight")}, extra      # every other key is a model parameter
        assert set(mk) - set(got) == set()


@pytest.mark.gpu
def test_original_format_checkpoints_load_into_the_packed_models(tmp_path):
    """An original-key, fp8-scaled Wan file and a BFL Flux file through `load_checkpoint_into` (converter = the family's
    table): every parameter of the packed model equals the converted (and dequantised) tensor bit for bit, nothing is
    missing, only the consumed `scale_weight` entries are left over."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import converters as CV, weights
    from apex_studio_amd.flux import FluxTransformer2DModel
    from apex_studio_amd.wan import WanTransformer3DModel
    (pw, pf), (wan, flux), (wan_cfg, flux_cfg), _ = _original_files(tmp_path)
    for path, sd, conv, cls, cfg in ((pw, wan, CV.WanKeyConverter, WanTransformer3DModel, wan_cfg),
                                     (pf, flux, CV.FluxKeyConverter, FluxTransformer2DModel, flux_cfg)):
        m = cls(**cfg, device=DEV, dtype=torch.bfloat16)
        m.pack()
        missing, unexpected = weights.load_checkpoint_into(m, [path])
        assert missing == [] and unexpected == [], (missing[:4], unexpected[:4])
        want = conv().convert(dict(sd), [k for k, _ in m.named_parameters()])
        got = m.state_dict()
        for k, v in want.items():
            if k.endswith("scale_weight"):
                continue
            if v.dtype in weights.FP8_DTYPES:
                v = OW.dequant(v, want[k[:-len("weight")] + "scale_weight"])
            assert torch.equal(got[k].cpu(), v.to(torch.bfloat16)), k


@pytest.mark.gpu
def test_fp8_scaled_expert_stays_fp8_in_hbm_and_matches_dequant_at_load(tmp_path):
    """SURVEY.md §8f-2, second half: a Kijai-keyed fp8-scaled Wan file loaded with `keep_fp8=True` keeps every
    block Linear as float8 + scale on the device (`ops.Fp8Weight`, fused q|k|v with per-row scales) and dequantises per call, as the
    reference's FPScaledLinear does (scaled_layer.py:390-552); no bf16 copy of those weights exists in the model, and the forward
    equals the dequantise-at-load model's bit for bit."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import ops, weights
    from apex_studio_amd.wan import WanTransformer3DModel
    from tests.golden.seeded import seeded
    (pw, _), (wan, _), (wan_cfg, _), _ = _original_files(tmp_path, fp8_all_block_linears=True)
    n_fp8 = sum(v.numel() for k, v in wan.items() if v.dtype == torch.float8_e4m3fn and v.dim() == 2)
    a = WanTransformer3DModel(**wan_cfg, device=DEV, dtype=torch.bfloat16)
    assert weights.load_checkpoint_into(a, [pw]) == ([], [])
    b = WanTransformer3DModel(**wan_cfg, device=DEV, dtype=torch.bfloat16)
    assert weights.load_checkpoint_into(b, [pw], keep_fp8=True) == ([], [])
    # what is resident: fp8 bytes (+ 2 per scale), and NO bf16 storage behind the block Linears
    lin = [p for n, p in b.named_parameters() if n.startswith("blocks.") and n.endswith(".weight") and p.dim() <= 2
           and ".norm" not in n and "scale_shift" not in n]
    assert lin and all(p.numel() == 0 for p in lin), [tuple(p.shape) for p in lin if p.numel()][:4]
    assert isinstance(b.blocks[0]._wqkv, ops.Fp8Weight) and b.blocks[0]._wqkv.scale.numel() == 3 * 128
    assert n_fp8 <= b._fp8_bytes <= n_fp8 + 8192          # + 2 bytes per scale value
    bf16_bytes_a = sum(p.numel() * 2 for n, p in a.named_parameters() if n.startswith("blocks."))
    bf16_bytes_b = sum(p.numel() * 2 for n, p in b.named_parameters() if n.startswith("blocks.")) + b._fp8_bytes
    assert bf16_bytes_b < 0.56 * bf16_bytes_a
    x, txt, t = seeded((1, 16, 3, 16, 24), 41).to(DEV), seeded((1, 20, 64), 42).to(DEV), torch.tensor([537.0], device=DEV)
    outs = []
    for m in (a, b, b):
        outs.append(m(hidden_states=x.to(torch.bfloat16), timestep=t, encoder_hidden_states=txt.to(torch.bfloat16), return_dict=False)[0].clone())
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().std()) > 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    with pytest.raises(apex_studio_amd.lib.ApexMIError):
        b._packed = False
        b.pack()
    from apex_studio_amd.flux import FluxTransformer2DModel
    with pytest.raises(NotImplementedError):
        weights.load_checkpoint_into(FluxTransformer2DModel(patch_size=1, in_channels=64, num_layers=1, num_single_layers=1,
                                                            attention_head_dim=128, num_attention_heads=1, joint_attention_dim=128,
                                                            pooled_projection_dim=64, device=DEV), [pw], keep_fp8=True)


@pytest.mark.gpu
def test_lightx2v_keyed_lora_merges_like_its_peft_twin():
    """A LoRA keyed like the lightx2v Wan files (`diffusion_model.blocks.N.self_attn.q.lora_down.weight`, alpha, diff
    vectors; the key handling is pinned to the reference in tests/test_converters.py) loaded into the HIP Wan model merges
    to the same weights, bit for bit, as the same adapter given in PEFT keys."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lora
    from apex_studio_amd.wan import WanTransformer3DModel
    from oracle import wan as OWan
    from tests.golden.seeded import spec_tensors, synthetic_state_dict
    cfg = dict(patch_size=(1, 2, 2), num_attention_heads=1, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=64,
               freq_dim=256, ffn_dim=256, num_layers=2, cross_attn_norm=True, eps=1e-6)
    sd_model = synthetic_state_dict(OWan.WanTransformer3DModel(**cfg), 9)
    r, spec = 4, {}
    for i in range(2):
        for a, n in (("self_attn", "q"), ("self_attn", "o"), ("cross_attn", "k"), ("cross_attn", "v")):
            m = f"diffusion_model.blocks.{i}.{a}.{n}"
            spec.update({m + ".lora_down.weight": (r, 128), m + ".lora_up.weight": (128, r), m + ".alpha": ()})
        spec.update({f"diffusion_model.blocks.{i}.ffn.0.lora_down.weight": (r, 128), f"diffusion_model.blocks.{i}.ffn.0.lora_up.weight": (256, r),
                     f"diffusion_model.blocks.{i}.ffn.2.lora_down.weight": (r, 256), f"diffusion_model.blocks.{i}.ffn.2.lora_up.weight": (128, r),
                     f"diffusion_model.blocks.{i}.cross_attn.k.diff_b": (128,), f"diffusion_model.blocks.{i}.cross_attn.norm_k.diff": (128,)})
    raw = spec_tensors(spec, 3000)
    twins = []
    for form in ("lightx2v", "peft"):
        m = WanTransformer3DModel(**cfg, device=DEV, dtype=torch.bfloat16)
        m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd_model.items()}, strict=True)
        m.pack()
        sd = raw if form == "lightx2v" else lora.convert_lora_state_dict(raw, "wan.base", [k for k, _ in m.named_parameters()])
        if form == "peft":      # alpha is folded into the converted factors already: drop the entries, or it would be folded twice
            assert all(k.startswith("blocks.") and ("lora_A" in k or "lora_B" in k or k.endswith(".alpha")) for k in sd), sorted(sd)[:4]
            sd = {k: v for k, v in sd.items() if not k.endswith(".alpha")}
        m.load_lora_adapter({k: v.clone() for k, v in sd.items()}, adapter_name="lx")
        twins.append({k: v.clone() for k, v in m.state_dict().items()})
    changed = [k for k in twins[0] if not torch.equal(twins[0][k].cpu().float(), sd_model[k].to(torch.bfloat16).float())]
    assert len(changed) == 12 and all(".attn" in k or ".ffn" in k for k in changed), changed
    for k in twins[0]:
        assert torch.equal(twins[0][k], twins[1][k]), k


@pytest.mark.gpu
def test_fp8_resident_expert_runs_a_lightx2v_keyed_lora_at_run_time(tmp_path):
    """the Wan-2.2 manifest ships fp8-scaled experts AND auto-applies lightning LoRAs
    (R/manifest/video/wan-2.2-a14b-text-to-video-1.0.0.v1.yml:39-51, :108-116); the reference serves that pair at run time,
    `base(x) + scale * B(A(x))` around FPScaledLinear (R/src/lora/manager.py:454-606, R/src/quantize/scaled_layer.py:496-549).
    A keep_fp8 model has no bf16 weight to merge into: its block Linears get the adapters' factors attached to their fp8
    records and `ops.gemm` applies them as one skinny GEMM + a K extended by the padded rank.  Checked against the oracle run
    on dequantised + merged f32 weights, against the dequantise-at-load model (which merges), through a scale change, and
    back to the plain fp8 forward bit for bit."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lib, lora, ops, weights
    from apex_studio_amd.wan import WanTransformer3DModel
    from oracle import layers as OL, lora as OLR, wan as OWan
    from tests.golden.seeded import seeded, spec_tensors
    (pw, _), _, (cfg, _), _ = _original_files(tmp_path, fp8_all_block_linears=True)
    a = WanTransformer3DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    assert weights.load_checkpoint_into(a, [pw]) == ([], [])
    b = WanTransformer3DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    assert weights.load_checkpoint_into(b, [pw], keep_fp8=True) == ([], [])
    base_sd = {k: v.float().cpu() for k, v in a.state_dict().items()}
    with pytest.raises(lib.ApexMIError, match="keep_fp8"):
        b.state_dict()
    r, spec = 4, {}
    for i in range(2):
        for at, n in (("self_attn", "q"), ("self_attn", "o"), ("cross_attn", "q"), ("cross_attn", "k"), ("cross_attn", "v")):
            m = f"diffusion_model.blocks.{i}.{at}.{n}"
            spec.update({m + ".lora_down.weight": (r, 128), m + ".lora_up.weight": (128, r), m + ".alpha": ()})
        spec.update({f"diffusion_model.blocks.{i}.ffn.0.lora_down.weight": (r, 128), f"diffusion_model.blocks.{i}.ffn.0.lora_up.weight": (256, r),
                     f"diffusion_model.blocks.{i}.ffn.2.lora_down.weight": (r, 256), f"diffusion_model.blocks.{i}.ffn.2.lora_up.weight": (128, r),
                     f"diffusion_model.blocks.{i}.cross_attn.k.diff_b": (128,)})
    raw = {k: (v * 0.3 if v.dim() == 2 else v) for k, v in spec_tensors(spec, 3100).items()}
    x, txt, t = seeded((1, 16, 3, 16, 24), 41).to(torch.bfloat16), seeded((1, 20, 64), 42).to(torch.bfloat16), torch.tensor([537.0])

    def fwd(m):
        out = m(hidden_states=x.to(DEV), timestep=t.to(DEV), encoder_hidden_states=txt.to(DEV), return_dict=False)[0]
        torch.cuda.synchronize()
        return out.float().cpu()
    plain = fwd(b)
    assert torch.equal(plain, fwd(a))
    a.load_lora_adapter({k: v.clone() for k, v in raw.items()}, adapter_name="lx")
    b.load_lora_adapter({k: v.clone() for k, v in raw.items()}, adapter_name="lx")
    assert b._lora_pad == 64 and b._fp8_bytes > 0
    rec = b.blocks[0]._wkv2                                   # fused k | v record: two adapters, block-diagonal up factors
    assert isinstance(rec, ops.Fp8Weight) and rec.lora_A.shape == (64, 128) and rec.lora_B.shape == (256, 64)
    assert float(rec.lora_B[:128, r:].abs().max()) == 0 and float(rec.lora_B[128:, :r].abs().max()) == 0
    assert float(rec.lora_A[2 * r:].abs().max()) == 0 and b.blocks[0]._wqkv.lora_A is not None
    mods = lora.split_modules(lora.convert_lora_state_dict(raw, "wan.base", list(base_sd)))
    assert len(mods) == 14 and "blocks.1.ffn.net.2" in mods

    def oracle(scale):
        orc = OWan.WanTransformer3DModel(**cfg).eval()
        sd = dict(base_sd)
        for m, d in mods.items():
            sd[m + ".weight"] = OLR.merged_weight(sd[m + ".weight"], [(d["A"].float(), d["B"].float(), scale)])
        orc.load_state_dict(sd, strict=True)
        with torch.no_grad():
            return orc(x.float(), t, txt.float(), policy=OL.BF16_STORAGE), orc(x.float(), t, txt.float())
    rel = lambda u, v: float((u - v).norm() / v.norm())          # noqa: E731
    for scale in (1.0, 0.5):
        if scale != 1.0:
            a.set_adapters("lx", scale)
            b.set_adapters("lx", scale)
        ref16, ref32 = oracle(scale)
        got_b, got_a = fwd(b), fwd(a)
        e_like, e_true, e_emul = rel(got_b, ref16), rel(got_b, ref32), rel(ref16, ref32)
        print(f"[fp8 + run-time LoRA, scale {scale}] vs the bf16-storage oracle {e_like:.2e}, vs fp32 {e_true:.2e} (emulation "
              f"{e_emul:.2e}); vs the dequantise-at-load model with MERGED weights {rel(got_b, got_a):.2e}; LoRA changed the output "
              f"by {rel(got_b, plain):.2e}")
        assert rel(got_b, plain) > 2e-2, "the adapter must matter for this test to mean anything"
        assert e_like < 6e-3 and e_true < 2 * e_emul + 2e-3 and rel(got_b, got_a) < 6e-3
        assert torch.equal(fwd(b), got_b), "deterministic"
    b.disable_lora()
    assert b._lora_pad == 0 and b.blocks[0]._wkv2.lora_A is None
    assert torch.equal(fwd(b), plain), "without adapters the resident-fp8 forward is the plain one again, bit for bit"
    b.enable_lora()
    b.delete_adapters("lx")
    assert torch.equal(fwd(b), plain) and b._lora_pad == 0
