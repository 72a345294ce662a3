"""Parity of every HIP op (called through the C-ABI via apex_studio_amd.ops) against the CPU oracle /
a plain PyTorch fp32 reference on the same seeded inputs.  Tolerances are stated per test: outputs are
stored in bf16 (8 bits of mantissa, ulp = 2^-8 relative), accumulation is f32."""
import math
import os

import pytest
import torch

from tests.conftest import measured

from oracle import layers as OL
from tests.golden.seeded import seeded

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from apex_studio_amd import ops
    return ops


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _check(out, ref, rel_tol, what, ulp=2.0):
    """rel L2 error, plus max-abs within `ulp` bf16 ulps of the largest reference magnitude."""
    out, ref = out.float().cpu(), ref.float().cpu()
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    rel = _rel(out, ref)
    mx = float((out - ref).abs().max())
    bound = ulp * 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    assert rel < rel_tol, f"{what}: rel L2 {rel:.3e} >= {rel_tol}"
    assert mx <= bound, f"{what}: max abs {mx:.3e} > {bound:.3e}"


def _bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 136, 192), (16, 64, 3072),
                                   (4096, 3072, 64), (1000, 768, 1024)])
def test_gemm_bias(M, N, K):
    ops = _ops()
    a, w, b = _bf(seeded((M, K), 1)), _bf(seeded((N, K), 2, scale=K ** -0.5)), _bf(seeded((N,), 3))
    out = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV))
    ref = a.float() @ w.float().T + b.float()
    _check(out, ref, 3e-3, f"gemm {M}x{N}x{K}")


def test_gemm_no_bias_and_asymmetric():
    # A = identity-like selector against an asymmetric W catches operand / output transposes
    ops = _ops()
    M = N = 128
    K = 128
    a = torch.eye(M, K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125.0
    out = ops.gemm(_bf(a).to(DEV), _bf(w).to(DEV))
    assert torch.equal(out.float().cpu(), _bf(w).float().T.contiguous())


def test_gemm_gelu_epilogue():
    ops = _ops()
    M, N, K = 300, 512, 256
    a, w, b = _bf(seeded((M, K), 4)), _bf(seeded((N, K), 5, scale=K ** -0.5)), _bf(seeded((N,), 6))
    out = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), epilogue="gelu")
    ref = torch.nn.functional.gelu(a.float() @ w.float().T + b.float(), approximate="tanh")
    _check(out, ref, 3e-3, "gemm+gelu")


def test_gemm_gate_residual_inplace_and_strided():
    ops = _ops()
    M, N, K = 260, 256, 320
    big = _bf(seeded((M, K + 64), 7)).to(DEV)
    a = big[:, 32:32 + K]                       # lda > K, 64-byte offset
    assert a.stride(0) == K + 64
    w, b = _bf(seeded((N, K), 8, scale=K ** -0.5)), _bf(seeded((N,), 9))
    gate = seeded((N,), 10).to(DEV)
    xbuf = _bf(seeded((M, N + 128), 11)).to(DEV)
    x = xbuf[:, 64:64 + N]                      # strided residual/output, updated in place
    x0 = x.float().cpu().clone()
    ops.gemm(a, w.to(DEV), b.to(DEV), out=x, epilogue="gate_res", gate=gate, residual=x)
    ref = x0 + gate.cpu() * (a.float().cpu() @ w.float().T + b.float())
    _check(x, ref, 3e-3, "gemm+gate_res")
    # columns outside the view untouched
    assert torch.equal(xbuf[:, :64].cpu(), _bf(seeded((M, N + 128), 11))[:, :64])


@pytest.fixture
def gemm_config():
    from apex_studio_amd import lib
    yield lambda v: lib.tune_set("gemm.config", v)
    lib.tune_set("gemm.config", 0)


@pytest.mark.parametrize("cfg", [1, 2, 3, 6, 7])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 520, 192), (1024, 1024, 1024), (77, 3072, 256)])
def test_gemm_every_tiling(cfg, M, N, K, gemm_config):
    """128x128, 256x256 and the 256x256 ping-pong schedule must agree with the fp32 reference."""
    ops = _ops()
    gemm_config(cfg)
    a, w, b = _bf(seeded((M, K), 1)), _bf(seeded((N, K), 2, scale=K ** -0.5)), _bf(seeded((N,), 3))
    gate, r = seeded((N,), 4), _bf(seeded((M, N), 5))
    ref = a.float() @ w.float().T + b.float()
    _check(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV)), ref, 3e-3, f"cfg{cfg} bias")
    _check(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), epilogue="gelu"),
           torch.nn.functional.gelu(ref, approximate="tanh"), 3e-3, f"cfg{cfg} gelu")
    x = r.to(DEV).clone()
    ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), out=x, epilogue="gate_res", gate=gate.to(DEV), residual=x)
    _check(x, r.float() + gate * ref, 3e-3, f"cfg{cfg} gate_res")


@pytest.mark.parametrize("cfg", [1, 2, 3, 6, 7])
def test_gemm_pingpong_race_screen(cfg, gemm_config):
    """Repeat a deep-K problem: a staging/barrier race shows up as run-to-run differences."""
    ops = _ops()
    gemm_config(cfg)
    M, N, K = 1024, 2048, 4096
    a = _bf(seeded((M, K), 11)).to(DEV)
    w = _bf(seeded((N, K), 12, scale=K ** -0.5)).to(DEV)
    ref = a.float() @ w.float().T
    first = ops.gemm(a, w)
    _check(first, ref, 3e-3, f"cfg{cfg} deep-K")
    for _ in range(10):
        assert torch.equal(ops.gemm(a, w), first), "non-deterministic result: LDS staging race"


@pytest.mark.parametrize("cfg", [1, 3, 6, 7])
def test_gemm_grouped(cfg, gemm_config):
    """img + txt streams in one launch, writing row ranges of one joint buffer."""
    ops = _ops()
    gemm_config(cfg)
    K, N, Mi, Mt = 512, 768, 700, 80
    ai, at = _bf(seeded((Mi, K), 1)), _bf(seeded((Mt, K), 2))
    wi, wt = _bf(seeded((N, K), 3, scale=K ** -0.5)), _bf(seeded((N, K), 4, scale=K ** -0.5))
    bi, bt = _bf(seeded((N,), 5)), _bf(seeded((N,), 6))
    out = torch.zeros(Mt + Mi, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_grouped([ai.to(DEV), at.to(DEV)], [wi.to(DEV), wt.to(DEV)], [bi.to(DEV), bt.to(DEV)],
                     [out[Mt:], out[:Mt]])
    _check(out[Mt:], ai.float() @ wi.float().T + bi.float(), 3e-3, "grouped img")
    _check(out[:Mt], at.float() @ wt.float().T + bt.float(), 3e-3, "grouped txt")
    gi, gt = seeded((N,), 7).to(DEV), seeded((N,), 8).to(DEV)
    x = _bf(seeded((Mt + Mi, N), 9)).to(DEV)
    x0 = x.float().cpu().clone()
    ops.gemm_grouped([ai.to(DEV), at.to(DEV)], [wi.to(DEV), wt.to(DEV)], [bi.to(DEV), bt.to(DEV)],
                     [x[Mt:], x[:Mt]], epilogue="gate_res", gate_list=[gi, gt], residual_list=[x[Mt:], x[:Mt]])
    _check(x[Mt:], x0[Mt:] + gi.cpu() * (ai.float() @ wi.float().T + bi.float()), 3e-3, "grouped img gate")
    _check(x[:Mt], x0[:Mt] + gt.cpu() * (at.float() @ wt.float().T + bt.float()), 3e-3, "grouped txt gate")


@pytest.mark.parametrize("cfg", [1, 3, 6, 7])
def test_gemm_grouped_mixed_n_and_epilogue(cfg, gemm_config):
    """QKV (bias) + MLP-up (gelu) of a single block: same input, different N, one launch."""
    ops = _ops()
    gemm_config(cfg)
    M, K, N1, N2 = 1100, 256, 768, 1024
    a = _bf(seeded((M, K), 1))
    w1, w2 = _bf(seeded((N1, K), 2, scale=K ** -0.5)), _bf(seeded((N2, K), 3, scale=K ** -0.5))
    b1, b2 = _bf(seeded((N1,), 4)), _bf(seeded((N2,), 5))
    o1 = torch.empty(M, N1, dtype=torch.bfloat16, device=DEV)
    cat = torch.zeros(M, 256 + N2, dtype=torch.bfloat16, device=DEV)
    ag = a.to(DEV)
    ops.gemm_grouped([ag, ag], [w1.to(DEV), w2.to(DEV)], [b1.to(DEV), b2.to(DEV)], [o1, cat[:, 256:]],
                     epilogue=["bias", "gelu"])
    _check(o1, a.float() @ w1.float().T + b1.float(), 3e-3, "grouped qkv")
    _check(cat[:, 256:], torch.nn.functional.gelu(a.float() @ w2.float().T + b2.float(), approximate="tanh"),
           3e-3, "grouped mlp")
    assert torch.equal(cat[:, :256].cpu(), torch.zeros(M, 256, dtype=torch.bfloat16))


def test_gemm_flux_shapes_against_gpu_fp32():
    ops = _ops()
    for (M, N, K) in [(4608, 9216, 3072), (4608, 3072, 15360), (512, 12288, 3072)]:
        a = _bf(seeded((M, K), 21)).to(DEV)
        w = _bf(seeded((N, K), 22, scale=K ** -0.5)).to(DEV)
        b = _bf(seeded((N,), 23)).to(DEV)
        out = ops.gemm(a, w, b)
        ref = a.float() @ w.float().T + b.float()
        _check(out, ref, 3e-3, f"gemm {M}x{N}x{K}")


def test_gemm_rejects_bad_shapes():
    ops = _ops()
    from apex_studio_amd.lib import ApexMIError
    a = torch.zeros(128, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ApexMIError):
        ops.gemm(a, a)  # K % 64 != 0


# ------------------------------------------------------------------------------------------- GEMV
@pytest.mark.parametrize("N,K,pre,post", [(256, 256, False, None), (3072, 256, False, "silu"),
                                          (18432, 3072, True, None), (1000, 768, False, "gelu")])
def test_gemv(N, K, pre, post):
    ops = _ops()
    w, b, x = _bf(seeded((N, K), 31, scale=K ** -0.5)), _bf(seeded((N,), 32)), seeded((1, K), 33)
    y = ops.gemv(w.to(DEV), x.to(DEV), b.to(DEV), pre_silu=pre, post=post)
    xi = torch.nn.functional.silu(x) if pre else x
    ref = xi @ w.float().T + b.float()
    if post == "silu":
        ref = torch.nn.functional.silu(ref)
    if post == "gelu":
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    assert torch.allclose(y.cpu(), ref, atol=2e-4, rtol=2e-4), float((y.cpu() - ref).abs().max())
    y2 = ops.gemv(w.to(DEV), x.to(DEV), b.to(DEV), out=y.clone(), pre_silu=pre, post=post, accum=True)
    assert torch.allclose(y2.cpu(), 2 * ref, atol=4e-4, rtol=4e-4)


@pytest.mark.parametrize("M,N,K", [(2, 70, 256), (5, 301, 768), (28, 517, 3072), (9, 64, 5120), (3, 40, 16384)])
@pytest.mark.parametrize("pre,post,accum", [(True, None, False), (False, "silu", False), (False, None, True)])
def test_gemv_rows_are_bit_identical_to_single_row_launches(M, N, K, pre, post, accum):
    """M > 1 takes the multi-row kernel (one pass over W per group of rows, x rows in LDS): the modulation table of a whole clip.
    Every row must equal the single-row launch on that x — same per-lane chunk order, same fmaf chain, same butterfly."""
    ops = _ops()
    w, b = _bf(seeded((N, K), 31, scale=K ** -0.5)).to(DEV), _bf(seeded((N,), 32)).to(DEV)
    x = seeded((M, K), 33).to(DEV)
    y0 = seeded((M, N), 34).to(DEV)
    kw = dict(pre_silu=pre, post=post, accum=accum)
    many = ops.gemv(w, x, b, out=y0.clone() if accum else None, **kw)
    for m in range(M):
        one = ops.gemv(w, x[m:m + 1], b, out=y0[m:m + 1].clone() if accum else None, **kw)
        assert torch.equal(one[0], many[m]), (m, float((one[0] - many[m]).abs().max()))
    # strided output rows (a column block of a wider table) and no bias
    wide = torch.zeros(M, N + 24, device=DEV)
    ops.gemv(w, x, None, out=wide[:, 8:8 + N], pre_silu=pre, post=post)
    assert torch.equal(wide[:, 8:8 + N], ops.gemv(w, x, None, pre_silu=pre, post=post)) and float(wide[:, :8].abs().sum()) == 0.0


# ------------------------------------------------------------------------------------ LN / modulate
@pytest.mark.parametrize("M,C", [(5, 256), (131, 3072), (64, 5120), (33, 3584)])
def test_ln_modulate(M, C):
    ops = _ops()
    x = _bf(seeded((M, C), 41) * 3 + 0.5)
    scale, shift = seeded((C,), 42) * 0.3, seeded((C,), 43) * 0.3
    out = ops.ln_modulate(x.to(DEV), scale.to(DEV), shift.to(DEV), eps=1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6) * (1 + scale) + shift
    _check(out, ref, 3e-3, f"ln_modulate {M}x{C}")
    out = ops.ln_modulate(x.to(DEV), eps=1e-6)
    _check(out, torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6), 3e-3, "plain LN")
    g, bt = _bf(1 + 0.1 * seeded((C,), 44)), _bf(0.1 * seeded((C,), 45))
    out = ops.ln_modulate(x.to(DEV), gamma=g.to(DEV), beta=bt.to(DEV), eps=1e-6)
    _check(out, torch.nn.functional.layer_norm(x.float(), (C,), g.float(), bt.float(), 1e-6), 3e-3, "affine LN")
    out = ops.ln_modulate(x.to(DEV), gamma=g.to(DEV), eps=1e-6, rms=True)
    n = OL.RMSNorm(C, 1e-6)
    with torch.no_grad():
        n.weight.copy_(g.float())
    _check(out, n(x.float()), 3e-3, "rmsnorm")


def test_ln_modulate_split_streams():
    ops = _ops()
    M, C, split = 150, 3072, 37
    x = _bf(seeded((M, C), 46) * 2)
    sc, sh, sc2, sh2 = (seeded((C,), 47 + i) * 0.3 for i in range(4))
    out = ops.ln_modulate(x.to(DEV), sc.to(DEV), sh.to(DEV), split=split, scale2=sc2.to(DEV), shift2=sh2.to(DEV))
    ln = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6)
    ref = torch.cat([ln[:split] * (1 + sc2) + sh2, ln[split:] * (1 + sc) + sh])
    _check(out, ref, 3e-3, "ln_modulate split")


def test_rmsnorm_matches_reference_golden(golden_dir):
    ops = _ops()
    c = torch.load(os.path.join(golden_dir, "efficiency_ops.pt"), weights_only=False)["rmsnorm_bf16"]
    x = c["x"][0]
    out = ops.ln_modulate(x.to(DEV), gamma=_bf(c["weight"]).to(DEV), eps=c["eps"], rms=True)
    # the reference rounds the normalisation factor to bf16 before the multiply (mod.py:31-33)
    assert torch.allclose(out.float().cpu(), c["out"][0].float(), atol=3e-2, rtol=2e-2)
    measured("efficiency_ops.rmsnorm_bf16.rel_vs_reference_run", _rel(out.cpu(), c["out"][0]), 7e-3)      # measured 3.3e-3 (round 6)


# ------------------------------------------------------------------------------------- qkv_prepare
def _qkv_ref(qkv, H, wq, wk, wq2, wk2, split, cos, sin):
    S = qkv.shape[0]
    dim = H * 128
    q, k, v = (qkv[:, i * dim:(i + 1) * dim].float().reshape(1, S, H, 128) for i in range(3))

    def rn(x, w):
        n = OL.RMSNorm(128, 1e-6)
        with torch.no_grad():
            n.weight.copy_(w.float())
        return n(x)

    def norm(x, w, w2):
        if split:
            return torch.cat([rn(x[:, :split], w2), rn(x[:, split:], w)], dim=1)
        return rn(x, w)

    q, k = norm(q, wq, wq2), norm(k, wk, wk2)
    q = OL.apply_rotary_emb(q, (cos, sin), sequence_dim=1)
    k = OL.apply_rotary_emb(k, (cos, sin), sequence_dim=1)
    return q[0].permute(1, 0, 2), k[0].permute(1, 0, 2), v[0].permute(1, 2, 0)  # [H,S,D],[H,S,D],[H,D,S]


@pytest.mark.parametrize("S,H,split", [(80, 2, 16), (200, 3, 0), (64, 24, 7), (131, 4, 0), (100, 8, 20)])
def test_qkv_prepare(S, H, split):
    ops = _ops()
    from apex_studio_amd import lib
    from oracle.flux import flux_pos_embed
    dim = H * 128
    qkv = _bf(seeded((S, 3 * dim), 51))
    ws = [_bf(1 + 0.1 * seeded((128,), 52 + i)) for i in range(4)]
    ids = torch.zeros(S, 3)
    ids[:, 1] = torch.arange(S) // 8
    ids[:, 2] = torch.arange(S) % 8
    cos, sin = flux_pos_embed(ids, (16, 56, 56))
    rope = ops.rope_table_axes(ids.to(DEV), (16, 56, 56))
    assert torch.allclose(rope[0].cpu(), cos, atol=1e-6) and torch.allclose(rope[1].cpu(), sin, atol=1e-6)
    skp = (S + 63) // 64 * 64
    qo = torch.empty(H, S, 128, dtype=torch.bfloat16, device=DEV)
    ko = torch.empty_like(qo)
    vt = torch.full((H, 128, skp), float("nan"), dtype=torch.bfloat16, device=DEV)
    g = qkv.to(DEV)
    ops.qkv_prepare(g[:, :dim], g[:, dim:2 * dim], g[:, 2 * dim:], H, qo, ko, vt,
                    wq=ws[0].to(DEV), wk=ws[1].to(DEV), wq2=ws[2].to(DEV), wk2=ws[3].to(DEV),
                    split=split, eps=1e-6, rope=rope, rope_mode=lib.ROPE_INTERLEAVED)
    rq, rk, rvt = _qkv_ref(qkv, H, ws[0], ws[1], ws[2], ws[3], split, cos, sin)
    _check(qo, rq, 3e-3, "q norm+rope")
    _check(ko, rk, 3e-3, "k norm+rope")
    assert torch.equal(vt[:, :, :S].float().cpu(), rvt), "V^T must be an exact transpose"
    assert torch.equal(vt[:, :, S:].float().cpu(), torch.zeros(H, 128, skp - S)), "V^T pad must be zero"


def test_qk_norm_rope_grouped_kernel_is_bit_identical():
    """The shipped launch (`qk.group=1`: four heads per lane group, V transpose in the same launch) against the four-head
    kernel alone (2) and the original one-head kernel + separate transpose (0): same arithmetic, identical bits."""
    ops = _ops()
    from apex_studio_amd import lib
    S, H = 333, 24
    dim = H * 128
    skp = (S + 63) // 64 * 64
    qkv = _bf(seeded((S, 3 * dim), 71)).to(DEV)
    ws = [_bf(1 + 0.1 * seeded((128,), 72 + i)).to(DEV) for i in range(4)]
    ang = seeded((S, 64), 76) * 3
    tables = {lib.ROPE_COMPLEX: torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(DEV),
              lib.ROPE_INTERLEAVED: torch.stack([ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1)]).contiguous().to(DEV)}
    try:
        for mode, table in tables.items():
            outs = []
            for grp in (0, 2, 1):
                lib.tune_set("qk.group", grp)
                qo = torch.empty(H, S, 128, dtype=torch.bfloat16, device=DEV)
                ko = torch.empty_like(qo)
                vt = torch.full((H, 128, skp), float("nan"), dtype=torch.bfloat16, device=DEV)
                ops.qkv_prepare(qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:], H, qo, ko, vt, wq=ws[0], wk=ws[1],
                                wq2=ws[2], wk2=ws[3], split=40, eps=1e-6, rope=table, rope_mode=mode)
                outs.append((qo, ko, vt))
            for o in outs[1:]:
                assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), mode
            assert torch.equal(outs[0][2][:, :, :S].cpu(), qkv[:, 2 * dim:].cpu().view(S, H, 128).permute(1, 2, 0))
    finally:
        lib.tune_set("qk.group", 1)


def test_rope_complex_mode_equals_interleaved():
    ops = _ops()
    from apex_studio_amd import lib
    S, H = 96, 4
    dim = H * 128
    qkv = _bf(seeded((S, 3 * dim), 61)).to(DEV)
    ang = seeded((S, 64), 62) * 3
    cplx = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous()          # [S, 64, 2]
    inter = torch.stack([ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1)]).contiguous()
    outs = []
    for table, mode in ((cplx, lib.ROPE_COMPLEX), (inter, lib.ROPE_INTERLEAVED)):
        qo = torch.empty(H, S, 128, dtype=torch.bfloat16, device=DEV)
        ko = torch.empty_like(qo)
        ops.qkv_prepare(qkv[:, :dim], qkv[:, dim:2 * dim], None, H, qo, ko, None, rope=table.to(DEV),
                        rope_mode=mode)
        outs.append((qo.clone(), ko.clone()))
    # same arithmetic up to fma association -> at most one bf16 ulp apart
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a.float(), b.float(), atol=0, rtol=2.0 ** -7)
    cos, sin = inter[0], inter[1]
    x = qkv[:, :dim].float().cpu().reshape(1, S, H, 128)
    ref = OL.apply_rotary_emb(x, (cos, sin), sequence_dim=1)[0].permute(1, 0, 2)
    _check(outs[0][0], ref, 3e-3, "complex rope q")


# --------------------------------------------------------------------------------------- attention
def test_attention_reference_goldens(golden_dir):
    """reference `sdpa` outputs (attention/functions.py:338-377) on the committed seeds."""
    ops = _ops()
    cases = torch.load(os.path.join(golden_dir, "attention_sdpa.pt"), weights_only=False)
    for c in cases:
        dt = torch.float32 if "float32" in c["dtype"] else torch.bfloat16
        q = seeded(c["q_shape"], c["seed"], dt)
        k = seeded(c["k_shape"], c["seed"] + 100, dt)
        v = seeded(c["k_shape"], c["seed"] + 200, dt)
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
        assert out.shape == c["out"].shape
        if dt == torch.float32:
            assert torch.allclose(out.cpu(), c["out"], atol=2e-5, rtol=2e-5), c["q_shape"]
        else:
            # bf16: P is rounded to bf16 before P V (as flash kernels do) -> ~1e-2 rel of |out|max
            _check(out, OL.sdpa(q.float(), k.float(), v.float()), 1e-2, f"attn {c['q_shape']}", ulp=3.0)
            assert torch.allclose(out.float().cpu(), c["out"].float(), atol=3e-2, rtol=3e-2)
            measured(f"attention_sdpa.bf16.{tuple(c['q_shape'])}.rel_vs_reference_run", _rel(out.cpu(), c["out"]), 6.5e-3)   # measured 2.3e-3 … 3.1e-3 over the cases
            if c["q_shape"][-1] == 128:      # the flash kernels (other head sizes run the generic kernel, whose probabilities stay f32)
                # like for like: the oracle rounding where the kernel rounds (bf16 P into P V, f32 row sums, bf16 store)
                measured(f"attention_sdpa.bf16.{tuple(c['q_shape'])}.like_for_like",
                         _rel(out.cpu(), OL.sdpa(q.float(), k.float(), v.float(), policy=OL.BF16_STORAGE).to(torch.bfloat16)), 5e-4)


def test_attention_verification_probe_shape():
    """B,H,S,D = 1,2,8,64 fp16 — the reference's backend self-test (functions.py:1999-2251)."""
    ops = _ops()
    q, k, v = (seeded((1, 2, 8, 64), 70 + i, torch.float16) for i in range(3))
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = OL.sdpa(q.float(), k.float(), v.float())
    assert torch.allclose(out.float().cpu(), ref, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("B,H,Sq,Sk", [(1, 2, 128, 64), (1, 3, 200, 333), (2, 2, 64, 1000), (1, 24, 1536, 1536)])
def test_attention_mfma_vs_oracle(B, H, Sq, Sk):
    ops = _ops()
    q = seeded((B, H, Sq, 128), 81, torch.bfloat16)
    k = seeded((B, H, Sk, 128), 82, torch.bfloat16)
    v = seeded((B, H, Sk, 128), 83, torch.bfloat16)
    # permuted [B,S,H,D]-backed views, as the flux processor passes them
    qv = q.permute(0, 2, 1, 3).contiguous().to(DEV).permute(0, 2, 1, 3)
    kv = k.permute(0, 2, 1, 3).contiguous().to(DEV).permute(0, 2, 1, 3)
    vv = v.permute(0, 2, 1, 3).contiguous().to(DEV).permute(0, 2, 1, 3)
    out = ops.attention(qv, kv, vv)
    assert out.permute(0, 2, 1, 3).is_contiguous()
    if Sq * Sk * H <= 4_000_000:
        ref = OL.sdpa(q.float(), k.float(), v.float())
    else:
        ref = OL.sdpa(q.float().to(DEV), k.float().to(DEV), v.float().to(DEV)).cpu()
    _check(out, ref, 1e-2, f"attention {B}x{H}x{Sq}x{Sk}", ulp=3.0)


@pytest.mark.parametrize("mfma", [16, 32])
@pytest.mark.parametrize("waves", [4, 5, 6, 7, 8])
def test_attention_workgroup_sizes(waves, mfma):
    """Both workgroup sizes of both MFMA kernels (32x32x16 shipped, 16x16x32 kept for A/B), ragged Sq / Sk."""
    from apex_studio_amd import lib
    ops = _ops()
    q = seeded((1, 2, 700, 128), 85, torch.bfloat16)
    k = seeded((1, 2, 333, 128), 86, torch.bfloat16)
    v = seeded((1, 2, 333, 128), 87, torch.bfloat16)
    lib.tune_set("attn.waves", waves)
    lib.tune_set("attn.mfma", mfma)
    try:
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    finally:
        lib.tune_set("attn.waves", 0)
        lib.tune_set("attn.mfma", 32)
    _check(out, OL.sdpa(q.float(), k.float(), v.float()), 1e-2, f"attention {waves} waves mfma{mfma}", ulp=3.0)


def test_attention_online_softmax_rescale_spike():
    """Force the running max to jump at a late KV tile (guide §5.4 rule 26)."""
    ops = _ops()
    H, S = 2, 512
    q = seeded((1, H, S, 128), 91, torch.bfloat16)
    k = seeded((1, H, S, 128), 92, torch.bfloat16)
    v = seeded((1, H, S, 128), 93, torch.bfloat16)
    k[0, :, 400] = (q[0, :, 17].float() * 4).to(torch.bfloat16)  # key 400 dominates query 17
    k[0, :, 130] = (q[0, :, 300].float() * 0.6).to(torch.bfloat16)  # modest jump (below the defer threshold)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = OL.sdpa(q.float(), k.float(), v.float())
    _check(out, ref, 1e-2, "attention spike", ulp=3.0)


def test_attention_flux_shape_properties():
    """Full Flux-1024 shape: linearity in V and row-stochasticity (V = 1 -> out = 1)."""
    ops = _ops()
    B, H, S = 1, 24, 4608
    q = seeded((B, H, S, 128), 101, torch.bfloat16).to(DEV)
    k = seeded((B, H, S, 128), 102, torch.bfloat16).to(DEV)
    ones = torch.ones(B, H, S, 128, dtype=torch.bfloat16, device=DEV)
    out = ops.attention(q, k, ones)
    assert torch.allclose(out.float(), torch.ones_like(out.float()), atol=8e-3)
    v = seeded((B, H, S, 128), 103, torch.bfloat16).to(DEV)
    o1 = ops.attention(q, k, v).float()
    o2 = ops.attention(q, k, (v.float() * 2).to(torch.bfloat16)).float()
    assert torch.allclose(o2, 2 * o1, atol=2e-2, rtol=2e-2)
    assert torch.equal(ops.attention(q, k, v).float(), o1), "attention must be deterministic (race screen)"
    # explicit-math f64 reference, one head at a time (independent of torch's SDPA backend choice)
    ref = torch.empty_like(o1)
    for h in range(H):
        sc = (q[0, h].double() @ k[0, h].double().T) / math.sqrt(128)
        ref[0, h] = (torch.softmax(sc, dim=-1) @ v[0, h].double()).float()
    _check(o1, ref, 1e-2, "attention flux-1024 shape", ulp=3.0)


# ------------------------------------------------------------------------------------------- misc
def test_timestep_embedding():
    ops = _ops()
    t = torch.tensor([0.0, 1.0, 718.75, 1000.0])
    out = ops.timestep_embedding(t.to(DEV), 256)
    ref = OL.get_timestep_embedding(t, 256, flip_sin_to_cos=True, downscale_freq_shift=0)
    assert torch.allclose(out.cpu(), ref, atol=2e-4), float((out.cpu() - ref).abs().max())


def test_euler_step_and_casts():
    ops = _ops()
    s, v = seeded((1, 4096, 64), 111), _bf(seeded((1, 4096, 64), 112))
    out = ops.euler_step(s.to(DEV), v.to(DEV), -0.0357)
    assert torch.allclose(out.cpu(), s + (-0.0357) * v.float(), atol=1e-6)
    sb = _bf(s)
    out = ops.euler_step(sb.to(DEV), v.to(DEV), -0.0357)
    # fused multiply-add vs separate ops: within one bf16 ulp
    assert torch.allclose(out.float().cpu(), sb.float() + (-0.0357) * v.float(), atol=1e-6, rtol=2.0 ** -7)
    assert torch.equal(ops.to_bf16(s.to(DEV)).cpu(), sb)
    assert torch.equal(ops.to_f32(sb.to(DEV)).cpu(), sb.float())


@pytest.mark.parametrize("shape", [(1, 1, 1024, 1024, 512), (2, 1, 1024, 1000, 384), (1, 2, 300, 2048, 256),
                                   (1, 1, 700, 1021, 384)])
def test_attention_materialised_wide_heads(shape):
    """Head dims the flash kernel does not cover (VAE mid blocks, C = 384 / 512): scores by the GEMM's f32
    epilogue, row softmax, P V by the GEMM.  Same bar as the flash kernel (P is rounded to bf16 before P V)."""
    ops = _ops()
    B, H, Sq, Sk, D = shape
    q = seeded((B, H, Sq, D), 181, torch.bfloat16)
    k = seeded((B, H, Sk, D), 182, torch.bfloat16)
    v = seeded((B, H, Sk, D), 183, torch.bfloat16)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = OL.sdpa(q.float().to(DEV), k.float().to(DEV), v.float().to(DEV)).cpu()
    _check(out, ref, 1e-2, f"attention materialised {shape}", ulp=3.0)
    out2 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert torch.equal(out.cpu(), out2.cpu())


def test_gemm_f32_epilogue():
    """APEXMI_EPI_BIAS_F32: float output of both tilings (scores of the materialised attention)."""
    from apex_studio_amd import lib
    import ctypes  # noqa: F401
    ops = _ops()
    for (M, N, K) in [(300, 264, 128), (1280, 1032, 512)]:
        a = seeded((M, K), 191, torch.bfloat16).to(DEV)
        w = seeded((N, K), 192, torch.bfloat16).to(DEV)
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        rc = lib.load().apexmi_gemm_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), None, out.data_ptr(),
                                         out.stride(0), M, N, K, 3, None, None, 0, torch.cuda.current_stream().cuda_stream)
        lib.check(rc, "gemm f32")
        ref = a.float() @ w.float().t()
        assert torch.isfinite(out).all()
        assert float((out - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), (M, N, K)


@pytest.mark.parametrize("c4", [0, 1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("shape", [(1, 2, 700, 333), (1, 4, 1536, 1536), (2, 3, 260, 1000), (1, 2, 256, 64)])
def test_attention_four_cluster_variant(shape, c4):
    """`attn.c4`: 4-cluster ping-pong kernel (K / V fragments burst-read into registers, halves one cluster apart).
    Same bar as the shipped kernel; bit-identical repeats screen the staggered LDS-DMA / barrier choreography."""
    from apex_studio_amd import lib
    ops = _ops()
    B, H, Sq, Sk = shape
    q = seeded((B, H, Sq, 128), 385, torch.bfloat16)
    k = seeded((B, H, Sk, 128), 386, torch.bfloat16)
    v = seeded((B, H, Sk, 128), 387, torch.bfloat16)
    lib.tune_set("attn.waves", 8)
    lib.tune_set("attn.c4", c4)
    lib.tune_set("attn.w64", 0)            # the variants of the 4-cluster kernel itself (the shipped main launch is attn.w64)
    try:
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
        out2 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV))
    finally:
        lib.tune_set("attn.waves", 0)
        lib.tune_set("attn.c4", 3)
        lib.tune_set("attn.w64", 1)
    _check(out, OL.sdpa(q.float(), k.float(), v.float()), 1e-2, f"attention c4 {shape}", ulp=3.0)
    measured(f"attention_c4.{shape}.like_for_like",       # measured 1e-8 .. 8.7e-5 (round 6)
             _rel(out.cpu(), OL.sdpa(q.float(), k.float(), v.float(), policy=OL.BF16_STORAGE).to(torch.bfloat16)), 3e-4)
    assert torch.equal(out.cpu(), out2.cpu())


@pytest.mark.parametrize("shape", [(1, 2, 256, 64), (1, 2, 256, 128), (1, 2, 256, 192), (1, 2, 256, 320), (1, 1, 64, 40),
                                   (1, 2, 700, 333), (2, 3, 260, 1000), (1, 4, 1536, 1536), (1, 3, 513, 4097)])
def test_attention_w64_kernel(shape):
    """`attn.w64` (shipped main launch): one wave per SIMD, 64 query rows per wave, the loop one generated asm statement
    (tools/gen_attn_w64.py).  One to five key tiles exercise every tail body of the generated loop (last / one / two more tiles),
    ragged key counts the masking of the last tile, 513 rows the clamped query rows.  Same bar as the 4-cluster kernel against
    the oracle, bit-identical repeats (race screen of the LDS ring / barrier choreography), and within one bf16 ulp on a few
    elements in 1e5 of the 4-cluster kernel (same rounding points, different f32 summation order of the row sums)."""
    from apex_studio_amd import lib
    ops = _ops()
    B, H, Sq, Sk = shape
    q = seeded((B, H, Sq, 128), 785, torch.bfloat16)
    k = seeded((B, H, Sk, 128), 786, torch.bfloat16)
    v = seeded((B, H, Sk, 128), 787, torch.bfloat16)
    lib.tune_set("attn.waves", 8)
    try:
        lib.tune_set("attn.w64", 1)
        lib.attn_w64_fallbacks()
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
        out2 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
        again = lib.attn_w64_fallbacks()
        lib.tune_set("attn.w64", 8)
        run = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
        lib.tune_set("attn.w64", 0)
        c4 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
    finally:
        lib.tune_set("attn.waves", 0)
        lib.tune_set("attn.w64", 1)
    # the shipped loop keeps tile 0's integer maximum: no workgroup needed its second pass, and the result is the running-maximum
    # loop's bit for bit (an integer shift of the maximum changes no mantissa)
    assert again == 0, again
    assert torch.equal(out.cpu(), run.cpu())
    _check(out, OL.sdpa(q.float(), k.float(), v.float()), 1e-2, f"attention w64 {shape}", ulp=3.0)
    # the assert that can fail: same inputs, the oracle rounding exactly where the kernel rounds -> the like-for-like bar (5e-4)
    measured(f"attention_w64.{shape}.like_for_like",      # measured 0 .. 1.1e-4 (round 6)
             _rel(out.cpu(), OL.sdpa(q.float(), k.float(), v.float(), policy=OL.BF16_STORAGE).to(torch.bfloat16)), 3e-4)
    assert torch.equal(out.cpu(), out2.cpu())
    frac = float((out != c4).float().mean())
    rel = float((out.float() - c4.float()).norm() / c4.float().norm())
    assert frac < 2e-3 and rel < 2e-4, (shape, frac, rel)


def test_attention_w64_running_max_rescale_and_flux_shape():
    """The w64 kernel's rare paths at full width: a key whose score towers over the running max late in the sequence forces the
    per-row rescale of O^T (through the AGPR accumulators), and the Flux launch geometry (24 heads x 4608, 432 workgroups)."""
    from apex_studio_amd import lib
    ops = _ops()
    H, S = 24, 4608
    q = seeded((1, H, S, 128), 811, torch.bfloat16)
    k = seeded((1, H, S, 128), 812, torch.bfloat16)
    v = seeded((1, H, S, 128), 813, torch.bfloat16)
    k[0, :, 3000] = q[0, :, 100] * 6.0          # row 100 gets a score far above everything seen in the first 46 tiles
    k[0, :, 4500] = q[0, :, 2000] * 8.0
    lib.tune_set("attn.w64", 1)
    lib.attn_w64_fallbacks()
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
    again = lib.attn_w64_fallbacks()
    try:
        lib.tune_set("attn.w64", 8)
        run = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
    finally:
        lib.tune_set("attn.w64", 1)
    rows = torch.cat([torch.arange(90, 110), torch.arange(1990, 2010), torch.arange(0, S, 257)])
    ref = OL.sdpa(q[:, :, rows].float(), k.float(), v.float())
    _check(out[:, :, rows], ref, 1e-2, "attention w64 rescale", ulp=3.0)
    assert torch.isfinite(out).all()
    # row 100's score stands ~90 binades over its first tile's best (row sum > 2^60), row 2000's ~125: exactly the workgroups
    # that hold those rows (one per head each) took the running-maximum pass and carry that loop's result bit for bit.  The other
    # rows see the two planted keys 10-25 binades above their first tile: no second pass, same rounding points, but the
    # exponent's f32 argument s c - m is rounded at another magnitude where the two loops hold different maxima -> a bf16 ulp on
    # a few elements
    assert again == 2 * H, again
    a, b = out.cpu(), run.cpu()
    for r0 in (0, 1792):
        assert torch.equal(a[:, :, r0:r0 + 256], b[:, :, r0:r0 + 256]), r0
    frac = float((a != b).float().mean())
    rel = _rel(a, b)
    assert frac < 2e-3 and rel < 2e-4, (frac, rel)
    measured("attention_w64.first_vs_running.peaked", rel, 2e-4)


def test_attention_w64_fallback_on_non_finite_and_uniform_rows():
    """The first-tile-maximum loop's check is `!(row sum <= 2^60)`: inf / NaN scores take the running-maximum pass, so a launch
    with poisoned rows returns what the running-maximum kernel returns (NaN rows included, nothing else touched); rows whose
    scores are all equal (q = 0) and rows far BELOW their first tile never trigger it."""
    from apex_studio_amd import lib
    ops = _ops()
    H, S = 3, 2048                          # 24 workgroups of 256 rows -> forced onto the w64 kernel through attn.waves
    q = seeded((1, H, S, 128), 821, torch.bfloat16)
    k = seeded((1, H, S, 128), 822, torch.bfloat16)
    v = seeded((1, H, S, 128), 823, torch.bfloat16)
    q[0, :, 300] = 0                        # uniform row
    k[0, :, :64] *= 6.0                     # the first tile holds every row's best score by far: later tiles sit far below
    q[0, 1, 700, 5] = float("inf")
    q[0, 2, 1500, 9] = float("nan")
    lib.tune_set("attn.waves", 8)
    try:
        lib.tune_set("attn.w64", 1)
        lib.attn_w64_fallbacks()
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
        again = lib.attn_w64_fallbacks()
        lib.tune_set("attn.w64", 8)
        run = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)).clone()
    finally:
        lib.tune_set("attn.waves", 0)
        lib.tune_set("attn.w64", 1)
    assert again == 2, again                # the two workgroups that hold the poisoned rows
    a, b = out.cpu().float(), run.cpu().float()
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))
    good = torch.ones(H, S, dtype=torch.bool)
    good[1, 700] = False
    good[2, 1500] = False
    assert torch.isfinite(a[0][good]).all() and torch.isnan(a[0, 1, 700]).all() and torch.isnan(a[0, 2, 1500]).all()
    qq = q.clone()
    qq[0, 1, 700, 5] = 0
    qq[0, 2, 1500, 9] = 0
    ref = OL.sdpa(qq.float(), k.float(), v.float())
    assert _rel(a[0][good], ref[0][good]) < 1e-2


@pytest.mark.parametrize("cfg", [1, 2, 3, 6, 7])
def test_gemm_random_shape_sweep(cfg, gemm_config):
    """Seeded sweep over ragged problems for every tiling: M from 1 row up, N any multiple of 8, K any multiple of 64,
    padded leading dimensions, every epilogue, guard columns/rows around the output checked for stray writes."""
    ops = _ops()
    gemm_config(cfg)
    rng = torch.Generator().manual_seed(1000 + cfg)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng))

    epis = ["bias", "gelu", "gelu_erf", "silu", "quick_gelu", "gate_res"]
    for case in range(14):
        M = [1, 7, 31, 129, 255, 257, 511, 513][case % 8] if case < 8 else ri(1, 1500)
        N = 8 * ri(1, 200)
        K = 64 * ri(1, 20)
        pa, pw, pc = 8 * ri(0, 3), 8 * ri(0, 3), 8 * ri(0, 3)
        epi = epis[case % len(epis)]
        abuf = _bf(seeded((M, K + pa), 7 * case + 1)).to(DEV)
        wbuf = _bf(seeded((N, K + pw), 7 * case + 2, scale=K ** -0.5)).to(DEV)
        a, w = abuf[:, :K], wbuf[:, :K]
        b = _bf(seeded((N,), 7 * case + 3)).to(DEV)
        cbuf = torch.full((M + 2, N + pc + 8), 7.0, dtype=torch.bfloat16, device=DEV)
        out = cbuf[1:M + 1, 8:8 + N]
        ref = a.float() @ w.float().T + b.float()
        if epi == "gate_res":
            gate, r = seeded((N,), 7 * case + 4).to(DEV), _bf(seeded((M, N), 7 * case + 5)).to(DEV)
            ops.gemm(a, w, b, out=out, epilogue=epi, gate=gate, residual=r)
            ref = r.float() + gate * ref
        else:
            ops.gemm(a, w, b, out=out, epilogue=epi)
            ref = {"bias": lambda x: x, "gelu": lambda x: torch.nn.functional.gelu(x, approximate="tanh"),
                   "gelu_erf": torch.nn.functional.gelu, "silu": torch.nn.functional.silu,
                   "quick_gelu": lambda x: x * torch.sigmoid(1.702 * x)}[epi](ref)
        _check(out, ref.cpu(), 3e-3, f"cfg{cfg} sweep {case}: M{M} N{N} K{K} {epi}")
        guard = cbuf.clone()
        guard[1:M + 1, 8:8 + N] = 7.0
        assert bool((guard == 7.0).all()), f"cfg{cfg} sweep {case}: write outside the output view (M{M} N{N} K{K})"


def test_attention_random_length_sweep():
    """Flash path (D = 128) over ragged query / key lengths and strided (fused-QKV) operands."""
    ops = _ops()
    rng = torch.Generator().manual_seed(77)
    for case in range(10):
        B, H = int(torch.randint(1, 3, (1,), generator=rng)), int(torch.randint(1, 5, (1,), generator=rng))
        Sq = int(torch.randint(1, 700, (1,), generator=rng))
        Sk = Sq if case % 2 == 0 else int(torch.randint(1, 900, (1,), generator=rng))
        q = seeded((B, Sq, H, 128), 300 + case, torch.bfloat16).to(DEV).permute(0, 2, 1, 3)       # [B,H,S,D] views
        kv = seeded((B, Sk, 2, H, 128), 400 + case, torch.bfloat16).to(DEV)
        k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
        out = ops.attention(q, k, v)
        ref = OL.sdpa(q.float(), k.float(), v.float())
        _check(out, ref.cpu(), 1e-2, f"attention sweep {case}: B{B} H{H} Sq{Sq} Sk{Sk}", ulp=3.0)


@pytest.mark.parametrize("H,S", [(24, 8448), (8, 8448 + 100), (33, 2304)])
def test_attention_tail_split_matches_unsplit_and_reference(H, S):
    """Launches whose last round would keep <= 1/4 of the CUs busy run it as 4 key ranges + a merge (`attn.split`):
    same result as the single launch up to f32 summation order (partials are f32 numerators + integer maxima, merged with
    exact power-of-two weights), and within the attention bar of the fp32 reference.  (24, 8448) is the QwenImage-Edit shape: 792 workgroups = 3 rounds + 24."""
    ops = _ops()
    from apex_studio_amd import lib
    q = seeded((1, H, S, 128), 501, torch.bfloat16).to(DEV)
    k = seeded((1, H, S, 128), 502, torch.bfloat16).to(DEV)
    v = seeded((1, H, S, 128), 503, torch.bfloat16).to(DEV)
    skp = (S + 63) // 64 * 64
    vt = torch.zeros(1, H, 128, skp, dtype=torch.bfloat16, device=DEV)
    vt[..., :S] = v.transpose(2, 3)
    nqb = (S + 255) // 256
    tail = (nqb * H) % 256
    assert 0 < tail <= 64, "shape must exercise the split"
    assert lib.load().apexmi_attn_prepared_workspace_bytes(1, H, S, S) == tail * 4 * 256 * (128 * 4 + 8)
    outs = {}
    try:
        for split in (1, 0):
            lib.tune_set("attn.split", split)
            o = torch.full((1, S, H, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.attention_prepared(q, k, vt, o, S)
            outs[split] = o.clone()
            o2 = torch.empty_like(o)
            ops.attention_prepared(q, k, vt, o2, S)
            assert torch.equal(o, o2)
    finally:
        lib.tune_set("attn.split", 1)
    assert torch.isfinite(outs[1]).all()
    d = (outs[1].float() - outs[0].float()).abs().max() / outs[0].float().abs().max()
    assert float(d) < 2.0 ** -7, float(d)
    changed = (outs[1] != outs[0]).any(dim=(0, 3))                       # [S, H]: only the tail q-blocks may differ
    assert int(changed.sum()) <= tail * 256
    rows = slice(S - 300, S)
    ref = OL.sdpa(q[:, :, rows].float(), k.float(), v.float()).permute(0, 2, 1, 3)
    _check(outs[1][:, rows], ref.cpu(), 1e-2, f"attention tail split H{H} S{S}", ulp=3.0)


def test_attention_operator_uses_the_tail_split():
    """The registered operator (`apexmi_attn_fwd`, strided [B,H,S,D] views) takes the same tail split as the fused path."""
    ops = _ops()
    from apex_studio_amd import lib
    H, S = 24, 8448
    qkv = seeded((1, S, 3, H, 128), 601, torch.bfloat16).to(DEV)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    try:
        lib.tune_set("attn.split", 0)
        a = ops.attention(q, k, v).clone()
        lib.tune_set("attn.split", 1)
        b = ops.attention(q, k, v).clone()
    finally:
        lib.tune_set("attn.split", 1)
    assert not torch.equal(a, b), "the split path was not taken"
    assert float((a.float() - b.float()).abs().max() / a.float().abs().max()) < 2.0 ** -7
    ref = OL.sdpa(q[:, :, -200:].float(), k.float(), v.float())
    _check(b[:, :, -200:], ref.cpu(), 1e-2, "operator tail split", ulp=3.0)


def test_gemm_grouped_tail_problem_on_small_tiling():
    """Grouped launch = image stream filling whole rounds (8 x 32 tiles = 256) + a small text stream: the text problem
    goes out on the 128x128 tiling (`gemm.tail`); both ways agree with the reference."""
    ops = _ops()
    from apex_studio_amd import lib
    K, N, Mi, Mt = 512, 8192, 2048, 256
    ai, at = _bf(seeded((Mi, K), 1)).to(DEV), _bf(seeded((Mt, K), 2)).to(DEV)
    wi, wt = _bf(seeded((N, K), 3, scale=K ** -0.5)).to(DEV), _bf(seeded((N, K), 4, scale=K ** -0.5)).to(DEV)
    bi, bt = _bf(seeded((N,), 5)).to(DEV), _bf(seeded((N,), 6)).to(DEV)
    outs = {}
    try:
        for tail in (1, 0):
            lib.tune_set("gemm.tail", tail)
            out = torch.full((Mt + Mi, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm_grouped([ai, at], [wi, wt], [bi, bt], [out[Mt:], out[:Mt]], epilogue="gelu")
            outs[tail] = out.clone()
    finally:
        lib.tune_set("gemm.tail", 1)
    ref = torch.cat([torch.nn.functional.gelu(at.float() @ wt.float().T + bt.float(), approximate="tanh"),
                     torch.nn.functional.gelu(ai.float() @ wi.float().T + bi.float(), approximate="tanh")])
    for tail in (1, 0):
        _check(outs[tail], ref.cpu(), 3e-3, f"grouped tail={tail}")
    assert torch.equal(outs[1][Mt:], outs[0][Mt:])          # the image rows come from the same kernel either way


def test_attention_backend_key_padding_masks():
    """`hip_mfma(q, k, v, attn_mask=...)` for the mask forms a padded prompt batch brings (the reference forwards
    `attn_mask=attention_mask` at every call site): bool keep-masks and additive 0 / -inf masks, [B, Sk] and [B, 1, 1, Sk],
    padded tails and holes, against torch's masked softmax in f32.  Query-varying masks still raise."""
    import torch.nn.functional as F
    from apex_studio_amd import attention_backend as ab
    from apex_studio_amd.lib import ApexMIError
    g = torch.Generator(device="cuda").manual_seed(17)
    B, H, Sq, Sk, D = 3, 4, 192, 320, 128
    q, k, v = (torch.randn(B, H, s, D, generator=g, device="cuda").to(torch.bfloat16) for s in (Sq, Sk, Sk))
    keep = torch.ones(B, Sk, dtype=torch.bool, device="cuda")
    keep[0, 200:] = False                    # padded tail
    keep[1, 17:40] = False                   # a hole
    keep[1, 300:] = False
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=keep[:, None, None, :])
    add = torch.zeros(B, 1, 1, Sk, device="cuda").masked_fill(~keep[:, None, None, :], float("-inf"))
    for m in (keep, keep[:, None, None, :], add, add.to(torch.bfloat16)):
        out = ab.hip_mfma(q, k, v, attn_mask=m)
        assert out.shape == ref.shape and out.dtype == q.dtype
        rel = float((out.float() - ref).norm() / ref.norm())
        assert rel < 4e-3, rel
    full = ab.hip_mfma(q, k, v, attn_mask=torch.ones(1, 1, 1, Sk, dtype=torch.bool, device="cuda"))
    assert torch.equal(full, ab.hip_mfma(q, k, v))
    with pytest.raises(ApexMIError):
        ab.hip_mfma(q, k, v, attn_mask=torch.ones(B, 1, Sq, Sk, dtype=torch.bool, device="cuda"))
    with pytest.raises(ApexMIError):
        ab.hip_mfma(q, k, v, attn_mask=torch.zeros(B, Sk, dtype=torch.bool, device="cuda"))


# ---- the 288 x 192 exact-fill tiling (round 5, gemm.hip gemm_bf16_x288_kernel) --------------------------------------------------
@pytest.fixture
def gemm_x288():
    from apex_studio_amd import lib
    yield lambda v: lib.tune_set("gemm.x288", v)
    lib.tune_set("gemm.x288", 0)


@pytest.mark.parametrize("M,N,K", [(4608, 3072, 15360), (4608, 3072, 256), (288, 192, 64), (289, 200, 128), (1000, 1040, 320),
                                   (75, 3072, 1024), (2000, 8, 64)])
def test_gemm_x288_tiling_is_bit_identical(M, N, K, gemm_x288):
    """Every output element is summed over K in the same order on both tilings (K-tiles in sequence, two 32-deep MFMAs each),
    so the 288 x 192 tiling must reproduce the shipped 256 x 256 launch BIT FOR BIT — every epilogue, ragged edges in M and N,
    in place on the residual — and match the fp32 reference."""
    ops = _ops()
    a, w, b = _bf(seeded((M, K), 1)).to(DEV), _bf(seeded((N, K), 2, scale=K ** -0.5)).to(DEV), _bf(seeded((N,), 3)).to(DEV)
    gate, r = seeded((N,), 4).to(DEV), _bf(seeded((M, N), 5)).to(DEV)
    got = {}
    for mode in (0, 2):
        gemm_x288(mode)
        x = r.clone()
        ops.gemm(a, w, b, out=x, epilogue="gate_res", gate=gate, residual=x)
        strided = torch.zeros(M, N + 24, dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, w, None, out=strided[:, 16:16 + N], epilogue="silu")
        got[mode] = (ops.gemm(a, w, b), ops.gemm(a, w, b, epilogue="gelu"), x, strided)
    for u, v, what in zip(got[0], got[2], ("bias", "gelu", "gate_res in place", "silu into a strided view, no bias")):
        assert torch.equal(u, v), f"288x192 differs from 256x256: {what} at {(M, N, K)}"
    ref = a.float() @ w.float().T + b.float()
    _check(got[2][0], ref, 3e-3, "x288 bias")
    _check(got[2][2], r.float() + gate * ref, 3e-3, "x288 gate_res")
    assert torch.equal(got[2][3][:, :16].cpu(), torch.zeros(M, 16, dtype=torch.bfloat16)) and \
        torch.equal(got[2][3][:, 16 + N:].cpu(), torch.zeros(M, 8, dtype=torch.bfloat16)), "stores outside the view"


def test_gemm_x288_grouped_and_race_screen(gemm_x288):
    """Two problems (different M, N and activation) in one 288 x 192 launch, and a deep-K problem repeated: a staging / barrier
    race (the counted vmcnt waits behind the duplicated pieces) shows up as run-to-run differences."""
    ops = _ops()
    K, Mi, Mt = 512, 700, 80
    ai, at = _bf(seeded((Mi, K), 1)).to(DEV), _bf(seeded((Mt, K), 2)).to(DEV)
    wi, wt = _bf(seeded((768, K), 3, scale=K ** -0.5)).to(DEV), _bf(seeded((1000, K), 4, scale=K ** -0.5)).to(DEV)
    bi, bt = _bf(seeded((768,), 5)).to(DEV), _bf(seeded((1000,), 6)).to(DEV)
    outs = {}
    for mode in (0, 2):
        gemm_x288(mode)
        oi, ot = torch.zeros(Mi, 768, dtype=torch.bfloat16, device=DEV), torch.zeros(Mt, 1000, dtype=torch.bfloat16, device=DEV)
        ops.gemm_grouped([ai, at], [wi, wt], [bi, bt], [oi, ot], epilogue=["bias", "gelu"])
        outs[mode] = (oi, ot)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    _check(outs[2][1], torch.nn.functional.gelu(at.float() @ wt.float().T + bt.float(), approximate="tanh"), 3e-3, "x288 grouped gelu")
    gemm_x288(2)
    a = _bf(seeded((1152, 4096), 11)).to(DEV)
    w = _bf(seeded((2048, 4096), 12, scale=4096 ** -0.5)).to(DEV)
    first = ops.gemm(a, w)
    _check(first, a.float() @ w.float().T, 3e-3, "x288 deep-K")
    for _ in range(10):
        assert torch.equal(ops.gemm(a, w), first), "non-deterministic result: LDS staging race"


def test_gemm_x288_auto_rule_and_shipped_default():
    """`gemm.x288`: 0 (SHIPPED — the exact-fill tiling measured 3.5 % slower on the very launch it was built for,
    profiles/r05_gemm_x288_ab.log) never uses it; 1 applies the rounds x tile-work rule: the single block's proj_out
    (4608 x 3072 x 15360: 216 tiles of 256 x 256, or 16 x 16 = 256 of 288 x 192) yes; launches that already fill their rounds
    (FF-up image stream, QKV + MLP, Wan's 23-round launches) no.  The result is the same bits either way."""
    from apex_studio_amd import lib
    ops = _ops()
    L = lib.load()
    assert L.apexmi_gemm_uses_x288(4608, 3072, 15360) == 0, "shipped default: off"
    a = _bf(seeded((4608, 15360), 1)).to(DEV)
    w = _bf(seeded((3072, 15360), 2, scale=15360 ** -0.5)).to(DEV)
    off = ops.gemm(a, w)
    lib.tune_set("gemm.x288", 1)
    try:
        assert L.apexmi_gemm_uses_x288(4608, 3072, 15360) == 1
        assert L.apexmi_gemm_uses_x288(4096, 12288, 3072) == 0       # FF-up, image stream: 3 exact rounds already
        assert L.apexmi_gemm_uses_x288(4608, 21504, 3072) == 0       # QKV + MLP-up: 5.9 rounds
        assert L.apexmi_gemm_uses_x288(75600, 5120, 5120) == 0       # Wan: 23.1 rounds
        assert L.apexmi_gemm_uses_x288(512, 3072, 3072) == 0         # under 1024 rows: the 128 x 128 tiling's business
        auto = ops.gemm(a, w)
    finally:
        lib.tune_set("gemm.x288", 0)
    assert torch.equal(auto, off)


def test_gemm_small_gate_res_launches_use_the_128_tiling():
    """`gemm.small_max` (round 5): a gate / residual launch of at most 112 tiles of 256 x 256 (the 512^2 geometry's proj_out,
    attention-out, FF-down) goes out as 128 x 128 tiles; same result as the forced 256 x 256 launch to f32 summation order,
    both within the usual bar of the fp32 reference; 144 tiles and more stay on 256 x 256 (bit-identical to forcing it)."""
    from apex_studio_amd import lib
    ops = _ops()
    for (M, N, K), small in (((1536, 3072, 1024), True), ((3072, 3072, 512), False)):
        a, w, b = _bf(seeded((M, K), 1)).to(DEV), _bf(seeded((N, K), 2, scale=K ** -0.5)).to(DEV), _bf(seeded((N,), 3)).to(DEV)
        gate, r = seeded((N,), 4).to(DEV), _bf(seeded((M, N), 5)).to(DEV)
        ref = r.float() + gate * (a.float() @ w.float().T + b.float())
        auto = ops.gemm(a, w, b, epilogue="gate_res", gate=gate, residual=r)
        lib.tune_set("gemm.small_max", 0)
        try:
            big = ops.gemm(a, w, b, epilogue="gate_res", gate=gate, residual=r)
        finally:
            lib.tune_set("gemm.small_max", 112)
        _check(auto, ref, 3e-3, f"small rule {M}x{N}")
        _check(big, ref, 3e-3, f"256x256 {M}x{N}")
        assert torch.equal(auto, big) != small or _rel(auto, big) < 1e-3, "the rule must (not) have changed the tiling"
        if not small:
            assert torch.equal(auto, big)


@pytest.mark.parametrize("epi", ["gelu", "silu", "quick_gelu"])
def test_gemm_epilogue_activations_on_every_bf16_code_point(epi):
    """the epilogue activations are x * rcp(1 + exp2(-z)) (v_exp_f32 + v_rcp_f32, ~1 ulp each) in EVERY kernel,
    the f32-storage verification mode included.  Pinned here on all 65 536 bf16 code points as the pre-activation value (one
    GEMM whose K-sum is exactly x): the bf16 output is within ONE bf16 ulp of the float64 definition rounded to bf16 (and equal
    to it on > 99 % of the points), the float output of the verification mode within 1e-6 relative (5e-7 of |x| absolute near the
    zero of the output), NaN stays NaN, +-inf and the saturated tails behave as the definition does."""
    ops = _ops()
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    x = bits.float()
    a = torch.zeros(65536, 64, dtype=torch.bfloat16)
    a[:, 0] = bits
    w = torch.zeros(8, 64, dtype=torch.bfloat16)
    w[:, 0] = 1.0
    xd = x.double()
    # 0.5 (1 + tanh(u)) = sigmoid(2 u) exactly; the tanh form loses the tail to cancellation even in float64
    ref = {"gelu": xd * torch.sigmoid(2 * 0.7978845608028654 * (xd + 0.044715 * xd ** 3)),
           "silu": xd * torch.sigmoid(xd), "quick_gelu": xd * torch.sigmoid(1.702 * xd)}[epi]
    fin = torch.isfinite(x)
    got16 = ops.gemm(a.to(DEV), w.to(DEV), None, epilogue=epi)[:, 0].float().cpu()
    got32 = ops.gemm(a.float().to(DEV), w.to(DEV), None, epilogue=epi)[:, 0].cpu()
    assert got32.dtype == torch.float32
    assert torch.isnan(got16[torch.isnan(x)]).all() and torch.isnan(got32[torch.isnan(x)]).all()
    # +inf -> +inf; -inf -> NaN for the x * sigmoid forms (inf * 0), as torch's own formulas give
    assert got16[x == float("inf")].item() == float("inf")
    r16 = ref.to(torch.bfloat16).float()
    ok = fin & torch.isfinite(r16) & ((x.abs() >= 2.0 ** -120) | (x == 0))      # subnormal results are the hardware's denormal mode's business
    # distance in bf16 ulps of the reference value
    ulp = torch.maximum(r16.abs(), torch.tensor(2.0 ** -100)).log2().floor().exp2() * 2.0 ** -7      # absolute 2^-107 below 2^-100
    d = ((got16 - r16).abs() / ulp)[ok]
    exact = float((d == 0).float().mean())
    print(f"[{epi}] bf16 output on {int(ok.sum())} finite code points: {100 * exact:.2f} % equal to the float64 definition rounded to bf16, "
          f"max distance {float(d.max()):.1f} bf16 ulp")
    assert float(d.max()) <= 1.0 and exact > 0.99
    err = (got32.double() - ref).abs()[ok]
    bound = 1e-6 * ref.abs()[ok] + 5e-7 * xd.abs()[ok].clamp_max(1e30) * (ref.abs()[ok] < 1e-3 * xd.abs()[ok]).double() + 1e-30
    worst = float((err / bound).max())
    print(f"[{epi}] float output of the verification mode: max error / (1e-6 |y| [+ 5e-7 |x| where |y| < 1e-3 |x|]) = {worst:.3f}")
    assert worst <= 1.0, worst


# ---- the 384 x 256 tiling (round 5, gemm.hip gemm_bf16_x384_kernel) ---------------------------------------------------------------
@pytest.fixture
def gemm_x384():
    from apex_studio_amd import lib
    yield lambda v: lib.tune_set("gemm.x384", v)
    lib.tune_set("gemm.x384", 1)


@pytest.mark.parametrize("M,N,K", [(4608, 3072, 3072), (384, 256, 64), (385, 264, 128), (1000, 1040, 320), (75, 3072, 1024),
                                   (2000, 8, 64), (768, 512, 15360)])
def test_gemm_x384_tiling_is_bit_identical(M, N, K, gemm_x384):
    """Same K order per output element as the 256 x 256 launch: the 384 x 256 tiling (192 accumulators, buffer-descriptor
    LDS-DMA with rows past M / N reading as zero) must reproduce it BIT FOR BIT — every epilogue, ragged edges, in place on the
    residual, strided output — and match the fp32 reference."""
    ops = _ops()
    a, w, b = _bf(seeded((M, K), 1)).to(DEV), _bf(seeded((N, K), 2, scale=K ** -0.5)).to(DEV), _bf(seeded((N,), 3)).to(DEV)
    gate, r = seeded((N,), 4).to(DEV), _bf(seeded((M, N), 5)).to(DEV)
    got = {}
    for mode in (0, 2):
        gemm_x384(mode)
        x = r.clone()
        ops.gemm(a, w, b, out=x, epilogue="gate_res", gate=gate, residual=x)
        strided = torch.zeros(M, N + 24, dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, w, None, out=strided[:, 16:16 + N], epilogue="silu")
        got[mode] = (ops.gemm(a, w, b), ops.gemm(a, w, b, epilogue="gelu"), x, strided)
    for u, v, what in zip(got[0], got[2], ("bias", "gelu", "gate_res in place", "silu into a strided view, no bias")):
        assert torch.equal(u, v), f"384x256 differs from 256x256: {what} at {(M, N, K)}"
    ref = a.float() @ w.float().T + b.float()
    _check(got[2][0], ref, 3e-3, "x384 bias")
    _check(got[2][2], r.float() + gate * ref, 3e-3, "x384 gate_res")
    assert torch.equal(got[2][3][:, :16].cpu(), torch.zeros(M, 16, dtype=torch.bfloat16)) and \
        torch.equal(got[2][3][:, 16 + N:].cpu(), torch.zeros(M, 8, dtype=torch.bfloat16)), "stores outside the view"


def test_gemm_x384_grouped_strided_operands_and_race_screen(gemm_x384):
    """Two problems in one 384 x 256 launch; activations read through a strided view (the single block's CAT buffer); a deep-K
    problem repeated (a race behind the counted vmcnt waits shows up as run-to-run differences)."""
    ops = _ops()
    K, Mi, Mt = 512, 700, 80
    big = _bf(seeded((Mi + Mt, K + 64), 1)).to(DEV)
    ai, at = big[Mt:, 32:32 + K], big[:Mt, 32:32 + K]
    wi, wt = _bf(seeded((768, K), 3, scale=K ** -0.5)).to(DEV), _bf(seeded((1000, K), 4, scale=K ** -0.5)).to(DEV)
    bi, bt = _bf(seeded((768,), 5)).to(DEV), _bf(seeded((1000,), 6)).to(DEV)
    outs = {}
    for mode in (0, 2):
        gemm_x384(mode)
        oi, ot = torch.zeros(Mi, 768, dtype=torch.bfloat16, device=DEV), torch.zeros(Mt, 1000, dtype=torch.bfloat16, device=DEV)
        ops.gemm_grouped([ai, at], [wi, wt], [bi, bt], [oi, ot], epilogue=["bias", "gelu"])
        outs[mode] = (oi, ot)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    _check(outs[2][1], torch.nn.functional.gelu(at.float() @ wt.float().T + bt.float(), approximate="tanh"), 3e-3, "x384 grouped gelu")
    gemm_x384(2)
    a = _bf(seeded((1152, 4096), 11)).to(DEV)
    w = _bf(seeded((2048, 4096), 12, scale=4096 ** -0.5)).to(DEV)
    first = ops.gemm(a, w)
    _check(first, a.float() @ w.float().T, 3e-3, "x384 deep-K")
    for _ in range(10):
        assert torch.equal(ops.gemm(a, w), first), "non-deterministic result: LDS staging race"
