"""Like-for-like parity: every HIP op against the CPU oracle evaluated WITH THE SAME ROUNDING POINTS.

BASELINE.json asks for outputs within 1e-3 (relative) of the CPU reference.  The GPU path stores activations in bf16,
whose rounding quantum alone is 2^-9 = 2e-3, so the comparison that can hold 1e-3 is the one SURVEY.md §7 ("tolerance vs
precision policy") prescribes: the oracle rounds to bf16 exactly where the kernels do (`oracle.layers.BF16_STORAGE`)
and everything in between is f32 on both sides.  What is then left is f32 summation order and 1-ulp differences of
exp2 / rsqrt / sin, which flip an occasional bf16 rounding: a relative perturbation d ahead of a bf16 rounding shows up
as sqrt(d * 2^-8) in relative L2 (flip probability d / ulp, flip size one ulp), i.e. about 1e-4 for the d = 3e-6 of a
K = 3072 f32 accumulation.  Bars asserted here: relative L2 <= 5e-4 per op (measured 1e-5 .. 1.3e-4) and no element
further from the oracle's value than one bf16 ulp + 2^-8 of the tensor's RMS.
"""
import math

import pytest
import torch

from oracle import layers as OL
from tests.golden.seeded import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"
POL = OL.BF16_STORAGE
REL = 5e-4


def _ops():
    from apex_studio_amd import ops
    return ops


def _bf(x):
    return x.to(torch.bfloat16)


def _like(out, ref_f32, what, rel_tol=REL, ulps=1.0):
    """`ref_f32` is the oracle value BEFORE its storage rounding; it is rounded here like the kernel's store."""
    out = out.float().cpu()
    ref = ref_f32.float().to(torch.bfloat16).float()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), what
    rel = float((out - ref).norm() / (ref.norm() + 1e-30))
    # per element: one bf16 spacing at |ref|, plus 2^-8 of the tensor's RMS for elements that are themselves a
    # cancellation of O(RMS) summands (an attention output near zero still carries the rounding flips of its ~1000
    # probabilities; a GELU output near zero the f32 noise of its pre-activation)
    rms = float(ref.pow(2).mean().sqrt())
    ulp = torch.exp2(torch.floor(torch.log2(torch.clamp(ref.abs(), min=1e-30))) - 7)
    worst = float(((out - ref).abs() / (ulp + rms * 2.0 ** -8)).max())
    nflip = int((out != ref).sum())
    print(f"[like-for-like] {what}: rel L2 {rel:.2e}, {nflip}/{ref.numel()} elements differ, worst {worst:.2f} ulp")
    assert rel <= rel_tol, f"{what}: rel L2 {rel:.3e} > {rel_tol}"
    assert worst <= ulps + 1e-3, f"{what}: {worst:.2f} bf16 ulp from the oracle"
    return rel


# ------------------------------------------------------------------------------------------------ GEMM epilogues
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (4608, 3072, 3072), (1000, 768, 15360)])
def test_gemm_epilogues_round_once(M, N, K):
    ops = _ops()
    a, w, b = _bf(seeded((M, K), 1)), _bf(seeded((N, K), 2, scale=K ** -0.5)), _bf(seeded((N,), 3))
    gate, res = seeded((N,), 4), _bf(seeded((M, N), 5))
    y = a.float() @ w.float().T + b.float()
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    _like(ops.gemm(ad, wd, bd), y, f"gemm bias {M}x{N}x{K}")
    _like(ops.gemm(ad, wd, bd, epilogue="gelu"), torch.nn.functional.gelu(y, approximate="tanh"), "gemm gelu-tanh")
    _like(ops.gemm(ad, wd, bd, epilogue="silu"), torch.nn.functional.silu(y), "gemm silu")
    x = res.to(DEV).clone()
    ops.gemm(ad, wd, bd, out=x, epilogue="gate_res", gate=gate.to(DEV), residual=x)
    _like(x, res.float() + gate * y, "gemm gate*y+residual (one rounding)")


def test_gemv_chain_is_f32():
    ops = _ops()
    K, N = 3072, 6 * 3072
    x = seeded((1, K), 11)
    w, b = _bf(seeded((N, K), 12, scale=K ** -0.5)), _bf(seeded((N,), 13))
    out = ops.gemv(w.to(DEV), x.to(DEV), b.to(DEV), pre_silu=True).cpu()
    ref = torch.nn.functional.silu(x) @ w.float().T + b.float()
    rel = float((out - ref).norm() / ref.norm())
    print(f"[like-for-like] gemv(silu(x)) f32: rel {rel:.2e}")
    assert rel < 2e-6


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,C", [(150, 3072), (64, 5120), (33, 3584), (40, 1024)])
def test_ln_modulate(M, C):
    ops = _ops()
    x = _bf(seeded((M, C), 21) * 2 + 0.3)
    sc, sh = seeded((C,), 22) * 0.3, seeded((C,), 23) * 0.3
    out = ops.ln_modulate(x.to(DEV), sc.to(DEV), sh.to(DEV))
    ref = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6) * (1 + sc) + sh
    _like(out, ref, f"ln_modulate {M}x{C}")
    g = _bf(1 + 0.1 * seeded((C,), 24))
    out = ops.ln_modulate(x.to(DEV), gamma=g.to(DEV), eps=1e-6, rms=True)
    n = OL.RMSNorm(C, 1e-6)
    with torch.no_grad():
        n.weight.copy_(g.float())
    _like(out, n(x.float()), f"rmsnorm {M}x{C}")


def test_qkv_prepare_norm_rope_one_rounding():
    ops = _ops()
    from apex_studio_amd import lib
    from oracle.flux import flux_pos_embed
    S, H, split = 333, 24, 40
    dim = H * 128
    qkv = _bf(seeded((S, 3 * dim), 51))
    ws = [_bf(1 + 0.1 * seeded((128,), 52 + i)) for i in range(4)]
    ids = torch.zeros(S, 3)
    ids[:, 1] = torch.arange(S) // 16
    ids[:, 2] = torch.arange(S) % 16
    cos, sin = flux_pos_embed(ids, (16, 56, 56))
    rope = ops.rope_table_axes(ids.to(DEV), (16, 56, 56))
    skp = (S + 63) // 64 * 64
    qo = torch.empty(H, S, 128, dtype=torch.bfloat16, device=DEV)
    ko = torch.empty_like(qo)
    vt = torch.zeros(H, 128, skp, dtype=torch.bfloat16, device=DEV)
    g = qkv.to(DEV)
    ops.qkv_prepare(g[:, :dim], g[:, dim:2 * dim], g[:, 2 * dim:], H, qo, ko, vt, wq=ws[0].to(DEV), wk=ws[1].to(DEV),
                    wq2=ws[2].to(DEV), wk2=ws[3].to(DEV), split=split, eps=1e-6, rope=rope, rope_mode=lib.ROPE_INTERLEAVED)

    def ref(x, w, w2):
        x = x.float().reshape(1, S, H, 128)
        n, n2 = torch.nn.RMSNorm(128, eps=1e-6), torch.nn.RMSNorm(128, eps=1e-6)
        with torch.no_grad():
            n.weight.copy_(w.float())
            n2.weight.copy_(w2.float())
            x = torch.cat([n2(x[:, :split]), n(x[:, split:])], dim=1)
        return OL.apply_rotary_emb(x, (cos, sin), sequence_dim=1)[0].permute(1, 0, 2)

    _like(qo, ref(qkv[:, :dim], ws[0], ws[2]), "q: per-head RMSNorm + RoPE")
    _like(ko, ref(qkv[:, dim:2 * dim], ws[1], ws[3]), "k: per-head RMSNorm + RoPE")


def test_timestep_embedding_arguments_are_the_references():
    """The frequency table comes from the host op sequence of the reference, so only sin / cos themselves can differ."""
    ops = _ops()
    t = torch.tensor([0.0, 1.0, 37.5, 500.0, 718.75, 999.0, 1000.0])
    for dim, scale, shift in ((256, 1.0, 0.0), (256, 1000.0, 0.0), (256, 1.0, 1.0)):
        tt = t / scale
        out = ops.timestep_embedding(tt.to(DEV), dim, scale=scale, downscale_freq_shift=shift).cpu()
        ref = OL.get_timestep_embedding(tt, dim, flip_sin_to_cos=True, downscale_freq_shift=shift, scale=scale)
        err = float((out - ref).abs().max())
        print(f"[like-for-like] timestep embedding dim {dim} scale {scale} shift {shift}: max abs {err:.2e}")
        assert err <= 4e-7, err


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,Sq,Sk", [(1, 2, 128, 64), (1, 3, 200, 333), (2, 2, 64, 1000), (1, 24, 1536, 1536),
                                       (1, 4, 2100, 512)])
def test_flash_attention_matches_oracle_rounding(B, H, Sq, Sk):
    ops = _ops()
    q, k, v = (seeded((B, H, S, 128), 81 + i, torch.bfloat16) for i, S in enumerate((Sq, Sk, Sk)))
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = OL.sdpa(q.float(), k.float(), v.float(), policy=POL)
    _like(out, ref, f"flash attention {B}x{H}x{Sq}x{Sk}")


def test_flash_attention_peaked_rows():
    """A late dominant key (running-max jump past the deferral threshold) and a modest one (below it)."""
    ops = _ops()
    H, S = 2, 512
    q, k, v = (seeded((1, H, S, 128), 91 + i, torch.bfloat16) for i in range(3))
    k[0, :, 400] = (q[0, :, 17].float() * 4).to(torch.bfloat16)
    k[0, :, 130] = (q[0, :, 300].float() * 0.6).to(torch.bfloat16)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    _like(out, OL.sdpa(q.float(), k.float(), v.float(), policy=POL), "flash attention, peaked rows")


def test_flash_attention_result_does_not_depend_on_the_schedule():
    """Integer running max => the bf16 rounding of P is the same for every workgroup height, MFMA shape and key split;
    only the f32 summation order differs, so the variants agree except for isolated 1-ulp flips."""
    from apex_studio_amd import lib
    ops = _ops()
    q, k, v = (seeded((1, 24, 2304, 128), 85 + i, torch.bfloat16).to(DEV) for i in range(3))
    outs = {}
    try:
        for waves, mfma in ((8, 32), (4, 32), (6, 32), (8, 16), (4, 16)):
            lib.tune_set("attn.waves", waves)
            lib.tune_set("attn.mfma", mfma)
            outs[(waves, mfma)] = ops.attention(q, k, v).float()
    finally:
        lib.tune_set("attn.waves", 0)
        lib.tune_set("attn.mfma", 32)
    base = outs[(8, 32)]
    for key, o in outs.items():
        frac = float((o != base).float().mean())
        rel = float((o - base).norm() / base.norm())
        print(f"[like-for-like] attention waves/mfma {key} vs (8, 32): {frac:.2e} of elements differ, rel {rel:.2e}")
        assert frac < 2e-3 and rel < 2e-4, (key, frac, rel)


def test_tail_split_rounds_where_the_single_launch_does():
    """QwenImage-Edit shape (792 workgroups = 3 rounds + 24): the key-split tail carries f32 partials, so the split
    launch differs from the single launch only by f32 summation order."""
    from apex_studio_amd import lib
    ops = _ops()
    H, S = 24, 8448
    q, k, v = (seeded((1, H, S, 128), 111 + i, torch.bfloat16).to(DEV) for i in range(3))
    try:
        lib.tune_set("attn.split", 0)
        single = ops.attention(q, k, v).float()
        lib.tune_set("attn.split", 1)
        split = ops.attention(q, k, v).float()
    finally:
        lib.tune_set("attn.split", 1)
    frac = float((single != split).float().mean())
    rel = float((single - split).norm() / single.norm())
    print(f"[like-for-like] tail split vs single launch: {frac:.2e} of elements differ, rel {rel:.2e}")
    assert frac < 2e-3 and rel < 2e-4, (frac, rel)
    rows = torch.arange(S - 256, S, 17)                         # rows of the split tail, against the oracle
    ref = OL.sdpa(q[0, -1:, rows].float().cpu(), k[0, -1:].float().cpu(), v[0, -1:].float().cpu(), policy=POL)
    _like(split[0, -1:, rows], ref, "tail rows vs oracle")


def test_materialised_attention_rounds_like_the_oracle():
    """VAE mid-block attention: GEMM -> row softmax (bf16 P after normalisation) -> GEMM."""
    ops = _ops()
    S, C = 1600, 384
    q, k, v = (_bf(seeded((1, 1, S, C), 121 + i)) for i in range(3))
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = OL.sdpa_materialized(q.float(), k.float(), v.float(), policy=POL)
    _like(out, ref, "materialised attention C=384")


# ------------------------------------------------------------------------------------------------ VAE ops
@pytest.mark.parametrize("cin,cout,T,H,W,k", [(96, 96, 3, 20, 24, (3, 3, 3)), (192, 96, 2, 16, 16, (3, 3, 3)),
                                              (384, 384, 2, 10, 12, (3, 3, 3)), (128, 128, 1, 32, 24, (1, 3, 3))])
def test_conv3d_bias_residual_round_once(cin, cout, T, H, W, k):
    """Implicit-GEMM convolution: f32 accumulation over all taps, bias and residual added in f32, ONE bf16 rounding."""
    import torch.nn.functional as F
    ops = _ops()
    x = _bf(seeded((T, H, W, cin), 1))
    w = _bf(seeded((cout, cin) + k, 2, scale=(cin * k[0] * k[1] * k[2]) ** -0.5))
    b, res = _bf(seeded((cout,), 3) * 0.1), _bf(seeded((T, H, W, cout), 4))
    wp = ops.pack_conv_weight(w.to(DEV))
    xin = F.pad(x.float().permute(3, 0, 1, 2)[None], (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0))
    ref = F.conv3d(xin, w.float(), b.float())[0].permute(1, 2, 3, 0)
    _like(ops.conv3d_cl(x.to(DEV), wp, b.to(DEV), k), ref, f"conv3d {cin}->{cout} {k}")
    _like(ops.conv3d_cl(x.to(DEV), wp, b.to(DEV), k, residual=res.to(DEV)), ref + res.float(), "conv3d + residual")


def test_vae_norms_round_once():
    import torch.nn.functional as F
    ops = _ops()
    for C in (96, 192, 384):
        x = _bf(seeded((2, 9, 11, C), 5) * 2)
        g = _bf(1 + 0.1 * seeded((C,), 6))
        for silu in (False, True):
            out = ops.rmsnorm_cl(x.to(DEV), g.to(DEV), silu=silu)
            ref = F.normalize(x.float(), dim=-1) * C ** 0.5 * g.float()           # WanRMS_norm.forward, model.py:216-222
            _like(out, F.silu(ref) if silu else ref, f"rmsnorm_cl C={C} silu={silu}")
    for C, hw in ((128, (40, 36)), (512, (33, 31))):
        x = _bf(seeded((1,) + hw + (C,), 21) * 2 + 0.3)
        g, b = _bf(1 + 0.1 * seeded((C,), 22)), _bf(0.1 * seeded((C,), 23))
        for silu in (False, True):
            y = ops.groupnorm_cl(x.to(DEV), g.to(DEV), b.to(DEV), silu=silu)
            ref = F.group_norm(x.float().permute(0, 3, 1, 2), 32, g.float(), b.float(), eps=1e-6).permute(0, 2, 3, 1)
            _like(y, F.silu(ref) if silu else ref, f"groupnorm_cl C={C} silu={silu}")
