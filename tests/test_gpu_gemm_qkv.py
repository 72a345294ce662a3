"""The q/k/v preparation fused into the QKV GEMM's epilogue (`apexmi_gemm_bf16_grouped_qkv`) against the two-pass path it
replaces (`apexmi_gemm_bf16_grouped` + `apexmi_qkv_prepare`, each of which has its own oracle parity tests): the projection is
rounded to bf16 where the two-pass path stores it and every sum runs in the same order, so the bar is BIT-IDENTITY — of the three
outputs kernel by kernel (joint img / txt streams, ragged row counts, the single block's QKV + MLP-up launch), and of a whole
Flux forward with the fusion on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def _rope(S, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    ang = torch.rand(S, 64, generator=g, device=DEV) * 6.283
    cos = ang.cos().repeat_interleave(2, dim=1)
    sin = ang.sin().repeat_interleave(2, dim=1)
    return torch.stack([cos, sin]).contiguous().float()


@pytest.fixture
def x384():
    """`gemm.x384`: 1 = the shipped rule, 2 = every launch on the 384 x 256 tiling, 0 = never; restored to 1."""
    from apex_studio_amd import lib as _l
    yield lambda v: _l.tune_set("gemm.x384", v)
    _l.tune_set("gemm.x384", 1)


@pytest.mark.parametrize("m_img,m_txt", [(1280, 256), (1096, 72), (1024, 0), (1091, 77), (1100, 3)])
def test_joint_streams_bit_identical_to_two_pass(m_img, m_txt):
    _joint_streams(m_img, m_txt)


@pytest.mark.parametrize("m_img,m_txt", [(1280, 256), (1152, 0), (1091, 77), (1100, 3), (700, 400)])
def test_joint_streams_on_the_384_tiling(m_img, m_txt, x384):
    """The same on the 384 x 256 tiling (round 5: `qkv_epilogue16<12, 192>`, V^T in two passes through the LDS): forced for the
    fused launch, with the two-pass reference left on the shipped 256 x 256 / 128 x 128 kernels."""
    _joint_streams(m_img, m_txt, x384)


def _joint_streams(m_img, m_txt, x384=None):
    from apex_studio_amd import lib as _l, ops
    H, K = 4, 512
    inner, S = H * 128, m_img + m_txt
    skp = (S + 63) // 64 * 64
    xs = [_rand((m_img, K), 1), _rand((max(m_txt, 1), K), 2)][:2 if m_txt else 1]
    ws = [_rand((3 * inner, K), 3, K ** -0.5), _rand((3 * inner, K), 4, K ** -0.5)][:len(xs)]
    bs = [_rand((3 * inner,), 5, 0.1), _rand((3 * inner,), 6, 0.1)][:len(xs)]
    nq = [_rand((128,), 7) * 0.2 + 1, _rand((128,), 8) * 0.2 + 1]
    nk = [_rand((128,), 9) * 0.2 + 1, _rand((128,), 10) * 0.2 + 1]
    rope = _rope(S, 11)
    row0 = [m_txt, 0][:len(xs)]
    # two-pass reference
    qkv = torch.empty(S, 3 * inner, device=DEV, dtype=torch.bfloat16)
    outs = [qkv[m_txt:], qkv[:m_txt]][:len(xs)]
    ops.gemm_grouped(xs, ws, bs, outs)
    q0, k0 = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt0 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    ops.qkv_prepare(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], H, q0, k0, vt0, wq=nq[0], wk=nk[0],
                    wq2=nq[1] if m_txt else None, wk2=nk[1] if m_txt else None, split=m_txt, eps=1e-6, rope=rope,
                    rope_mode=_l.ROPE_INTERLEAVED)
    # fused
    assert ops.qkv_fusable(xs, ws, row0, H)
    q1, k1 = (torch.full((H, S, 128), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt1 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    if x384 is not None:
        x384(2)
    ops.gemm_grouped_qkv(xs, ws, bs, [None] * len(xs), "bias", [1] * len(xs), nq[:len(xs)], nk[:len(xs)], row0, H, 1e-6, rope,
                         q1, k1, vt1)
    if x384 is not None:
        x384(1)
    torch.cuda.synchronize()
    assert torch.equal(q1, q0), int((q1 != q0).sum())
    assert torch.equal(k1, k0), int((k1 != k0).sum())
    assert torch.equal(vt1, vt0), int((vt1 != vt0).sum())


def test_single_block_launch_with_mlp_up_bit_identical():
    _single_block(4, 512, 1160, 2048)


def test_single_block_launch_on_the_384_tiling(x384):
    _single_block(4, 512, 1160, 2048, x384)


def test_flux_single_block_shape_takes_the_384_tiling_and_is_bit_identical(x384):
    """The real launch: M 4608 = 12 x 384, QKV (N 9216, 24 heads) + MLP-up (N 12288, GELU), K 3072 — 1008 tiles of 384 x 256 under the
    shipped rule (`gemm.x384` = 1) — against the two-pass path, and against the same fused launch kept on 256 x 256."""
    a = _single_block(24, 3072, 4608, 12288)
    x384(0)
    b = _single_block(24, 3072, 4608, 12288)
    x384(1)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def _single_block(H, K, S, mlp, x384=None):
    from apex_studio_amd import lib as _l, ops
    inner = H * 128
    skp = (S + 63) // 64 * 64
    x = _rand((S, K), 21)
    wqkv, wmlp = _rand((3 * inner, K), 22, K ** -0.5), _rand((mlp, K), 23, K ** -0.5)
    bqkv, bmlp = _rand((3 * inner,), 24, 0.1), _rand((mlp,), 25, 0.1)
    nq, nk = _rand((128,), 26) * 0.2 + 1, _rand((128,), 27) * 0.2 + 1
    rope = _rope(S, 28)
    qkv = torch.empty(S, 3 * inner, device=DEV, dtype=torch.bfloat16)
    up0 = torch.empty(S, mlp, device=DEV, dtype=torch.bfloat16)
    ops.gemm_grouped([x, x], [wqkv, wmlp], [bqkv, bmlp], [qkv, up0], epilogue=["bias", "gelu"])
    q0, k0 = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt0 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    ops.qkv_prepare(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], H, q0, k0, vt0, wq=nq, wk=nk, split=0, eps=1e-6,
                    rope=rope, rope_mode=_l.ROPE_INTERLEAVED)
    q1, k1 = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt1 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    up1 = torch.empty(S, mlp, device=DEV, dtype=torch.bfloat16)
    if x384 is not None:
        x384(2)
    ops.gemm_grouped_qkv([x, x], [wqkv, wmlp], [bqkv, bmlp], [None, up1], ["bias", "gelu"], [1, 0], [nq, None], [nk, None], [0, 0],
                         H, 1e-6, rope, q1, k1, vt1)
    if x384 is not None:
        x384(1)
    torch.cuda.synchronize()
    assert torch.equal(up1, up0) and torch.equal(q1, q0) and torch.equal(k1, k0) and torch.equal(vt1, vt0)
    return q1, k1, vt1, up1


def test_refused_when_the_tiling_has_no_fused_epilogue():
    from apex_studio_amd import lib as _l, ops
    x, w = _rand((256, 512), 1), _rand((1536, 512), 2)
    assert not ops.qkv_fusable([x], [w], [0], 4)                       # under 1024 rows: the 128x128 tiling
    big = _rand((1024, 512), 3)
    assert ops.qkv_fusable([big], [w], [0], 4)
    _l.tune_set("gemm.config", 3)
    try:
        assert not ops.qkv_fusable([big], [w], [0], 4)
        q = torch.empty(4, 1024, 128, device=DEV, dtype=torch.bfloat16)
        with pytest.raises(RuntimeError):
            ops.gemm_grouped_qkv([big], [w], [None], [None], "bias", [1], [None], [None], [0], 4, 1e-6, _rope(1024, 4), q, q.clone(),
                                 torch.zeros(4, 128, 1024, device=DEV, dtype=torch.bfloat16))
    finally:
        _l.tune_set("gemm.config", 0)


def test_flux_forward_identical_with_and_without_the_fusion():
    from apex_studio_amd.flux import FluxTransformer2DModel
    from tests.golden.seeded import seeded, synthetic_state_dict
    from oracle import flux as OF
    cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=4,
               joint_attention_dim=256, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    h2 = w2 = 32
    s_txt = 72
    m = FluxTransformer2DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    sd = synthetic_state_dict(m, 5)              # the HIP model's own parameter names are the diffusers names
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    inp = dict(hidden_states=seeded((1, h2 * w2, 64), 31).to(DEV).to(torch.bfloat16),
               encoder_hidden_states=seeded((1, s_txt, 256), 32).to(DEV).to(torch.bfloat16),
               pooled_projections=seeded((1, 64), 33).to(DEV).to(torch.bfloat16), timestep=torch.tensor([0.5], device=DEV),
               guidance=torch.tensor([4.0], device=DEV), img_ids=OF.latent_image_ids(h2, w2).to(DEV),
               txt_ids=torch.zeros(s_txt, 3, device=DEV))
    launched = []
    from apex_studio_amd import ops
    orig = ops.gemm_grouped_qkv
    ops.gemm_grouped_qkv = lambda *a, **k: (launched.append(1), orig(*a, **k))[1]
    try:
        m.fuse_qkv = True
        a = m(return_dict=False, **inp)[0].clone()
        n_fused = len(launched)
        m.fuse_qkv = False
        b = m(return_dict=False, **inp)[0].clone()
    finally:
        ops.gemm_grouped_qkv = orig
    torch.cuda.synchronize()
    assert n_fused == 4 and len(launched) == 4          # 2 joint + 2 single blocks took the fused launch, none when off
    assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0
    assert torch.equal(a, b)


@pytest.mark.parametrize("tiling", [256, 384])
def test_fused_epilogue_race_screen(tiling, x384):
    """(tiling 384: the same screen on the 384 x 256 tiling — its V^T leaves in two LDS passes, one per M-half.)
    The fused epilogue reuses the GEMM's staging LDS (cross-wave sums, the V tile) right after the main loop, whose two
    M-halves run a barrier apart: 60 back-to-back launches at a Flux-like shape (K deep enough for the ping-pong to be in steady
    state, unaligned text stream) must all give the same bits."""
    from apex_studio_amd import ops
    H, K = 6, 3072
    inner = H * 128
    m_img, m_txt = 2048, 77
    S = m_img + m_txt
    skp = (S + 63) // 64 * 64
    xs = [_rand((m_img, K), 31), _rand((m_txt, K), 32)]
    ws = [_rand((3 * inner, K), 33, K ** -0.5), _rand((3 * inner, K), 34, K ** -0.5)]
    bs = [_rand((3 * inner,), 35, 0.1), _rand((3 * inner,), 36, 0.1)]
    nq = [_rand((128,), 37) * 0.2 + 1, _rand((128,), 38) * 0.2 + 1]
    nk = [_rand((128,), 39) * 0.2 + 1, _rand((128,), 40) * 0.2 + 1]
    rope = _rope(S, 41)
    ref = None
    x384(2 if tiling == 384 else 0)
    for it in range(60):
        q, k = (torch.full((H, S, 128), 3.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
        vt = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
        ops.gemm_grouped_qkv(xs, ws, bs, [None, None], "bias", [1, 1], nq, nk, [m_txt, 0], H, 1e-6, rope, q, k, vt)
        cur = (q, k, vt)
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, ref)), it
    x384(1)
    assert torch.isfinite(ref[0].float()).all() and torch.isfinite(ref[2].float()).all()


@pytest.mark.parametrize("H,S,mode", [(24, 333, "interleaved"), (24, 200, "complex"), (40, 130, "interleaved"), (40, 64, "none")])
def test_wan_rms_rope_rows_bit_identical_to_three_passes(H, S, mode):
    """`apexmi_qk_rms_rope_rows` (Wan: RMSNorm across all heads of q and k, RoPE, layout, V^T in one pass) against the three
    passes it replaces — ln_modulate(rms) in place on q, on k, then qkv_prepare — and its q-only form (cross-attention)."""
    from apex_studio_amd import lib as _l, ops
    dim = H * 128
    skp = (S + 63) // 64 * 64
    qkv = _rand((S, 3 * dim), 51)
    wq, wk = _rand((dim,), 52) * 0.2 + 1, _rand((dim,), 53) * 0.2 + 1
    rm = {"interleaved": _l.ROPE_INTERLEAVED, "complex": _l.ROPE_COMPLEX, "none": _l.ROPE_NONE}[mode]
    rope = None
    if mode == "interleaved":
        rope = _rope(S, 54)
    elif mode == "complex":
        g = torch.Generator(device=DEV).manual_seed(55)
        ang = torch.rand(S, 64, generator=g, device=DEV) * 6.283
        rope = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().float()        # [S, 64, 2]
    ref = qkv.clone()
    q_in, k_in, v_in = ref[:, :dim], ref[:, dim:2 * dim], ref[:, 2 * dim:]
    ops.ln_modulate(q_in, gamma=wq, out=q_in, eps=1e-6, rms=True)
    ops.ln_modulate(k_in, gamma=wk, out=k_in, eps=1e-6, rms=True)
    q0, k0 = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt0 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    ops.qkv_prepare(q_in, k_in, v_in, H, q0, k0, vt0, rope=rope, rope_mode=rm)
    q1, k1 = (torch.full((H, S, 128), 5.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt1 = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
    x = qkv.clone()
    ops.qk_rms_rope_rows(x[:, :dim], x[:, dim:2 * dim], x[:, 2 * dim:], H, q1, k1, vt1, wq=wq, wk=wk, eps=1e-6, rope=rope,
                         rope_mode=rm)
    torch.cuda.synchronize()
    assert torch.equal(x, qkv), "the projection itself is not modified"
    assert torch.equal(q1, q0) and torch.equal(k1, k0) and torch.equal(vt1, vt0)
    q2 = torch.empty_like(q1)
    ops.qk_rms_rope_rows(x[:, :dim], None, None, H, q2, None, None, wq=wq, eps=1e-6, rope=rope, rope_mode=rm)
    assert torch.equal(q2, q0)


def test_wan_forward_identical_with_and_without_the_fused_rows_pass():
    from apex_studio_amd.wan import WanTransformer3DModel
    from tests.golden.seeded import seeded, synthetic_state_dict
    cfg = dict(patch_size=(1, 2, 2), num_attention_heads=24, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=256,
               freq_dim=256, ffn_dim=2048, num_layers=2, cross_attn_norm=True, eps=1e-6, rope_max_seq_len=1024)
    m = WanTransformer3DModel(**cfg, device=DEV, dtype=torch.bfloat16)
    m.load_state_dict({k: v.to(torch.bfloat16) for k, v in synthetic_state_dict(m, 13).items()}, strict=True)
    inp = dict(hidden_states=seeded((1, 16, 3, 16, 20), 61).to(DEV).to(torch.bfloat16), timestep=torch.tensor([700.0], device=DEV),
               encoder_hidden_states=seeded((1, 40, 256), 62).to(DEV).to(torch.bfloat16))
    calls = []
    from apex_studio_amd import ops
    orig = ops.qk_rms_rope_rows
    ops.qk_rms_rope_rows = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        m.fuse_qkv = True
        a = m(return_dict=False, **inp)[0].clone()
        n = len(calls)
        m.fuse_qkv = False
        b = m(return_dict=False, **inp)[0].clone()
    finally:
        ops.qk_rms_rope_rows = orig
    assert n == 4 and len(calls) == 4                    # 2 blocks x (self + cross query side)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def test_compact_rope_pairs_table_is_bit_identical_and_optional():
    """Round 6: the fused epilogue reads the rotary table from its compact [2, S, 64] copy when the table's entries come in equal
    pairs (`ops.rope_pairs`, `apexmi_rope_pairs` + `apexmi_gemm_bf16_grouped_qkv_pairs`): same outputs bit for bit as with the full
    table, on both tilings; a table that is NOT pair-duplicated has no compact copy and takes the full-table path."""
    from apex_studio_amd import lib as _l, ops
    H, K, S = 4, 512, 1536
    x, w, b = _rand((S, K), 1), _rand((3 * H * 128, K), 3, K ** -0.5), _rand((3 * H * 128,), 5, 0.1)
    nq, nk = _rand((128,), 7) * 0.2 + 1, _rand((128,), 9) * 0.2 + 1
    rope = _rope(S, 11)
    pairs = ops.rope_pairs(rope)
    assert pairs is not None and pairs.shape == (2, S, 64) and torch.equal(pairs, rope[:, :, ::2])
    assert ops.rope_pairs(rope) is pairs, "made once per table"
    odd = rope.clone()
    odd[0, :, 3] += 0.5
    assert ops.rope_pairs(odd) is None
    outs = {}
    try:
        for tiling in (1, 2):
            _l.tune_set("gemm.x384", tiling)
            for on in (True, False):
                ops.rope_pairs_enabled = on
                q, k = (torch.full((H, S, 128), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
                vt = torch.zeros(H, 128, S, device=DEV, dtype=torch.bfloat16)
                ops.gemm_grouped_qkv([x], [w], [b], [None], "bias", [1], [nq], [nk], [0], H, 1e-6, rope, q, k, vt)
                outs[(tiling, on)] = (q, k, vt)
    finally:
        ops.rope_pairs_enabled = True
        _l.tune_set("gemm.x384", 1)
    torch.cuda.synchronize()
    ref = outs[(1, False)]
    for key, o in outs.items():
        assert all(torch.equal(a, b_) for a, b_ in zip(o, ref)), key
    q, k = (torch.full((H, S, 128), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    vt = torch.zeros(H, 128, S, device=DEV, dtype=torch.bfloat16)
    ops.gemm_grouped_qkv([x], [w], [b], [None], "bias", [1], [nq], [nk], [0], H, 1e-6, odd, q, k, vt)   # full-table path, other values
    assert not torch.equal(q, ref[0])
