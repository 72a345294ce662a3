"""Teacher-forced stage parity (test infrastructure).

A free-running bf16 forward cannot stay within 1e-3 (relative L2) of ANY other bf16-storage evaluation for more than
a few kernels: a relative discrepancy d that reaches a bf16 store flips a fraction d / ulp of the roundings by one
ulp each, i.e. comes out as sqrt(d * ulp) in relative L2 (ulp ~ 3e-3) — 1e-5 becomes 2e-4, then 8e-4, 1.5e-3 ... and
the chain saturates at the bf16 noise floor of ~3e-3 after four or five storage points whatever the kernels do
(tools/parity_trace.py prints that staircase).  What CAN be checked at 1e-3 and below is every storage point on its
own: the oracle's forward is run once with every `Policy.r` call recorded (each one is a kernel output of the HIP
path); the HIP model then runs with a hook after every op that (1) compares the buffers the op wrote with the
oracle's values for that storage point and (2) overwrites them with the oracle's values ("teacher forcing"), so the
next kernel starts from bit-identical inputs.  Every kernel of the forward, in its real launch configuration inside the
model (grouped launches, joint buffers, in-place epilogues, strided views), is thereby compared like for like.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch

from oracle import layers as OL

OPS = ("gemm", "gemm_grouped", "ln_modulate", "qkv_prepare", "attention_prepared")


class TracePolicy(OL.Policy):
    """Storage policy that keeps every storage point, in call order: rounded to bf16 (the production policy), or — with
    emulate_bf16=False — the fp32 values themselves, for the f32-storage verification mode (tests/test_gpu_f32_storage.py)."""

    def __init__(self, emulate_bf16: bool = True):
        super().__init__(emulate_bf16)
        self.points: List[torch.Tensor] = []

    def r(self, x):
        out = x.to(torch.bfloat16).to(torch.float32) if self.emulate_bf16 else x
        self.points.append(out)
        return out


class Cursor:
    """Hands out the oracle's storage points in order, `take(n)` at a time."""

    def __init__(self, points: Sequence[torch.Tensor]):
        self.p, self.i = list(points), 0

    def take(self, n: int) -> List[torch.Tensor]:
        out = self.p[self.i:self.i + n]
        assert len(out) == n, "oracle trace shorter than the plan"
        self.i += n
        return out

    def done(self) -> bool:
        return self.i == len(self.p)


# a stage = (op name, [(label, view(ws) -> tensor the op wrote, oracle value in that layout)])
Stage = Tuple[str, List[Tuple[str, Callable, torch.Tensor]]]


def run_forced(ops_mod, model, plan: List[Stage], call: Callable[[], torch.Tensor], force: bool = True):
    """Run `call()` (one model forward) with the comparison / forcing hook installed.  Returns (output, report) with
    report = [(stage index, op, label, rel L2, elements that differ)]."""
    report = []
    it = iter(plan)
    orig = {n: getattr(ops_mod, n) for n in OPS}

    def wrap(name):
        def f(*a, **k):
            out = orig[name](*a, **k)
            try:
                op, pairs = next(it)
            except StopIteration:
                raise AssertionError(f"model launched more ops than the plan lists (at {name})")
            assert op == name, f"plan expects {op}, model launched {name} (stage {len(report)})"
            if pairs:
                torch.cuda.synchronize()
                ws = next(iter(model._ws.values()))
                for label, view, ref in pairs:
                    got = view(ws)
                    want = ref.to(device=got.device, dtype=got.dtype)
                    assert got.shape == want.shape, (label, got.shape, want.shape)
                    g, w = got.float(), want.float()
                    rel = float((g - w).norm() / (w.norm() + 1e-30))
                    report.append((len(report), name, label, rel, int((g != w).sum()), w.numel()))
                    if force:
                        got.copy_(want)
            return out
        return f

    for n in OPS:
        setattr(ops_mod, n, wrap(n))
    # the [S, 3 dim] QKV projection is a storage point of the plan: keep the two-pass path (the fused epilogue is asserted
    # bit-identical to it in tests/test_gpu_gemm_qkv.py, kernel by kernel and over a whole forward)
    fused = getattr(model, "fuse_qkv", None)
    if fused is not None:
        model.fuse_qkv = False
    try:
        out = call()
        torch.cuda.synchronize()
    finally:
        for n in OPS:
            setattr(ops_mod, n, orig[n])
        if fused is not None:
            model.fuse_qkv = fused
    leftover = sum(1 for _ in it)
    assert leftover == 0, f"{leftover} planned stages never ran"
    return out, report


def print_report(tag: str, report) -> Tuple[float, float]:
    worst = max(r[3] for r in report)
    mean = sum(r[3] for r in report) / len(report)
    for i, op, label, rel, nd, n in report:
        print(f"[stage {tag}] {i:3d} {op:18s} {label:28s} rel {rel:.2e}  differ {nd}/{n}")
    print(f"[stage {tag}] {len(report)} storage points: worst rel L2 {worst:.2e}, mean {mean:.2e}")
    return worst, mean


VAE_OPS = ("conv3d_cl", "conv3d_cl_norm", "conv3d_cl_act", "conv3d_cl_tstrided", "conv2d_cl_down2", "tanh_clamp", "rmsnorm_cl", "groupnorm_cl", "gemm",
           "attention", "attention_framecausal", "add", "group_mean")


def _oracle_rows(o: torch.Tensor) -> torch.Tensor:
    """An oracle storage point of a VAE ([1, C, T, H, W], [bt, C, H, W] or an attention output [bt, 1, HW, C]) as
    channels-last rows [positions, C], in the order the HIP tiles store them."""
    if o.dim() == 5:
        return o.permute(0, 2, 3, 4, 1).reshape(-1, o.shape[1])
    if o.dim() == 4 and o.shape[1] == 1:
        return o.reshape(-1, o.shape[-1])
    if o.dim() == 4:
        return o.permute(0, 2, 3, 1).reshape(-1, o.shape[1])
    if o.dim() == 3:                               # [b, HW, C] (Flux mid-block attention)
        return o.reshape(-1, o.shape[-1])
    raise AssertionError(f"unexpected oracle storage point {tuple(o.shape)}")


def run_forced_vae(ops_mod, points: Sequence[torch.Tensor], call: Callable[[], torch.Tensor], force: bool = True,
                   lookahead: int = 6):
    """Teacher forcing for the VAE decoders: their HIP classes write every oracle storage point with one op (the
    layout-only ops — frame interleave, pixel shuffle, channel repeat — write no new values), in ALMOST the oracle's
    order: a convolution that also emits the RMS norm of its output (`conv3d_cl_norm`) produces the consumer's norm
    before, e.g., the consumer's shortcut conv.  So each op output is matched against the next few unconsumed oracle
    points of its shape (the right one is ~1e-5 away, a wrong one ~1), compared, and overwritten.  A conv1 whose raw
    output lives only in registers (`want_raw=False`) is checked through its norm; its oracle point is ticked off."""
    report = []
    rows = [_oracle_rows(p) for p in points]
    used = [False] * len(rows)
    orig = {n: getattr(ops_mod, n) for n in VAE_OPS}

    def take(name, out, label=""):
        base = out.permute(0, 2, 1, 3) if name in ("attention", "attention_framecausal") else out   # [B,H,S,D] views of [B,S,H,D]
        assert base.is_contiguous(), name
        got = base.reshape(-1, base.shape[-1])
        torch.cuda.synchronize()
        best, seen = None, 0
        for idx in range(len(rows)):
            if used[idx]:
                continue
            seen += 1
            if seen > lookahead:
                break
            ref = rows[idx]
            if ref.shape[0] != got.shape[0] or ref.shape[1] > got.shape[1]:
                continue
            want = ref.to(device=got.device, dtype=got.dtype)
            g, w = got[:, :ref.shape[1]].float(), want.float()
            rel = float((g - w).norm() / (w.norm() + 1e-30))
            if best is None or rel < best[1]:
                best = (idx, rel, int((g != w).sum()), w.numel(), want)
        assert best is not None, f"no oracle storage point of shape {tuple(got.shape)} among the next {lookahead} ({name})"
        idx, rel, nd, n, want = best
        used[idx] = True
        report.append((len(report), name, f"{label}{tuple(out.shape)} -> point {idx}", rel, nd, n))
        if force:
            got[:, :want.shape[1]].copy_(want)
        return idx

    def wrap(name):
        def f(*a, **k):
            before = len(report)
            out = orig[name](*a, **k)
            if name == "conv3d_cl_norm" and len(report) > before:
                return out          # not fusable: the wrapper ran conv3d_cl + rmsnorm_cl, both hooked already
            if name == "conv3d_cl_norm":
                raw, normed = out
                if raw is not None:
                    take(name, raw, "raw ")
                j = take(name, normed, "norm ")
                if raw is None:        # the conv output itself was never stored: tick its point off (it precedes its norm)
                    prev = [i for i in range(j) if not used[i] and rows[i].shape == rows[j].shape]
                    assert prev, "the register-only conv output has no oracle point"
                    used[prev[-1]] = True
            else:
                take(name, out)
            return out
        return f

    for n in VAE_OPS:
        setattr(ops_mod, n, wrap(n))
    try:
        out = call()
        torch.cuda.synchronize()
    finally:
        for n in VAE_OPS:
            setattr(ops_mod, n, orig[n])
    left = [i for i, u in enumerate(used) if not u]
    assert not left, f"oracle storage points never produced by a HIP op: {left}"
    return out, report


STAGE_TOL = 5e-4


def assert_stages(tag: str, report, out, po) -> None:
    """The bar of north_star, per storage point: every kernel output within STAGE_TOL (relative L2) of the oracle's
    value given bit-identical inputs, and the model output (one kernel after the last forced point) as well."""
    print_report(tag, report)
    o, p = out.float().cpu(), po.float().cpu()
    e_out = float((o - p).norm() / p.norm())
    print(f"[stage {tag}] model output after the last forced point: rel {e_out:.2e}")
    bad = [r for r in report if r[3] > STAGE_TOL]
    assert not bad, f"storage points beyond {STAGE_TOL}: {[(r[2], r[3]) for r in bad]}"
    assert e_out <= STAGE_TOL, e_out


# ---- plans ------------------------------------------------------------------------------------------------------------
def _heads_from_bshd(v):     # oracle [1, S, H, 128] -> [H, S, 128]
    return v[0].permute(1, 0, 2).contiguous()


def flux_plan(points, cfg, s_txt: int) -> List[Stage]:
    """Op sequence of apex_studio_amd.flux._forward_one against the r() order of oracle.flux."""
    dim = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    T = s_txt
    c = Cursor(points)
    plan: List[Stage] = []
    x0, c0 = c.take(2)
    plan.append(("gemm", [("x_embedder", lambda ws: ws.X[T:], x0[0])]))
    plan.append(("gemm", [("context_embedder", lambda ws: ws.X[:T], c0[0])]))
    for b in range(cfg["num_layers"]):
        nx, nc, q, k, v, cq, ck, cv, qr, kr, o, xa, n2, ffh, xf, ca, c2, cffh, cf = c.take(19)
        t = f"d{b} "
        plan.append(("ln_modulate", [(t + "ln1 img", lambda ws: ws.XN[T:], nx[0]), (t + "ln1 txt", lambda ws: ws.XN[:T], nc[0])]))
        plan.append(("gemm_grouped", [
            (t + "q img", lambda ws: ws.QKV[T:, :dim], q[0].flatten(1)), (t + "k img", lambda ws: ws.QKV[T:, dim:2 * dim], k[0].flatten(1)),
            (t + "v img", lambda ws: ws.QKV[T:, 2 * dim:], v[0].flatten(1)), (t + "q txt", lambda ws: ws.QKV[:T, :dim], cq[0].flatten(1)),
            (t + "k txt", lambda ws: ws.QKV[:T, dim:2 * dim], ck[0].flatten(1)), (t + "v txt", lambda ws: ws.QKV[:T, 2 * dim:], cv[0].flatten(1))]))
        plan.append(("qkv_prepare", [(t + "q norm+rope", lambda ws: ws.Q[0], _heads_from_bshd(qr)),
                                     (t + "k norm+rope", lambda ws: ws.K[0], _heads_from_bshd(kr))]))
        plan.append(("attention_prepared", [(t + "attention", lambda ws: ws.CAT[:, :dim], o[0])]))
        plan.append(("gemm_grouped", [(t + "x + g*attn img", lambda ws: ws.X[T:], xa[0]), (t + "x + g*attn txt", lambda ws: ws.X[:T], ca[0])]))
        plan.append(("ln_modulate", [(t + "ln2 img", lambda ws: ws.XN[T:], n2[0]), (t + "ln2 txt", lambda ws: ws.XN[:T], c2[0])]))
        plan.append(("gemm_grouped", [(t + "gelu(ff up) img", lambda ws: ws.FFH[T:], ffh[0]), (t + "gelu(ff up) txt", lambda ws: ws.FFH[:T], cffh[0])]))
        plan.append(("gemm_grouped", [(t + "x + g*ff img", lambda ws: ws.X[T:], xf[0]), (t + "x + g*ff txt", lambda ws: ws.X[:T], cf[0])]))
    for b in range(cfg["num_single_layers"]):
        nh, mlp, q, k, v, qr, kr, o, h = c.take(9)
        t = f"s{b} "
        plan.append(("ln_modulate", [(t + "ln", lambda ws: ws.XN, nh[0])]))
        plan.append(("gemm_grouped", [
            (t + "q", lambda ws: ws.QKV[:, :dim], q[0].flatten(1)), (t + "k", lambda ws: ws.QKV[:, dim:2 * dim], k[0].flatten(1)),
            (t + "v", lambda ws: ws.QKV[:, 2 * dim:], v[0].flatten(1)), (t + "gelu(mlp)", lambda ws: ws.CAT[:, dim:], mlp[0])]))
        plan.append(("qkv_prepare", [(t + "q norm+rope", lambda ws: ws.Q[0], _heads_from_bshd(qr)),
                                     (t + "k norm+rope", lambda ws: ws.K[0], _heads_from_bshd(kr))]))
        plan.append(("attention_prepared", [(t + "attention", lambda ws: ws.CAT[:, :dim], o[0])]))
        plan.append(("gemm", [(t + "x + g*proj_out", lambda ws: ws.X, h[0])]))
    no, po = c.take(2)
    plan.append(("ln_modulate", [("norm_out", lambda ws: ws.XN[T:], no[0])]))
    plan.append(("gemm", []))          # proj_out writes a fresh tensor: compared as the model output
    assert c.done(), "oracle trace longer than the plan"
    return plan, po


def qwen_plan(points, cfg, s_txt: int):
    """apex_studio_amd.qwenimage._forward_one against oracle.qwenimage."""
    dim = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    T = s_txt
    c = Cursor(points)
    plan: List[Stage] = []
    img_in, txtn, txt_in = c.take(3)
    plan.append(("gemm", [("img_in", lambda ws: ws.X[T:], img_in[0])]))
    plan.append(("ln_modulate", [("txt_norm", lambda ws: ws.TXTN, txtn[0])]))
    plan.append(("gemm", [("txt_in", lambda ws: ws.X[:T], txt_in[0])]))
    for b in range(cfg["num_layers"]):
        im, tm, tq, iq, tk, ik, tv, iv, qr, kr, o, xa, ta, in2, iffh, xf, tn2, tffh, tf = c.take(19)
        t = f"b{b} "
        plan.append(("ln_modulate", [(t + "ln1 img", lambda ws: ws.XN[T:], im[0]), (t + "ln1 txt", lambda ws: ws.XN[:T], tm[0])]))
        plan.append(("gemm_grouped", [
            (t + "q img", lambda ws: ws.QKV[T:, :dim], iq[0].flatten(1)), (t + "k img", lambda ws: ws.QKV[T:, dim:2 * dim], ik[0].flatten(1)),
            (t + "v img", lambda ws: ws.QKV[T:, 2 * dim:], iv[0].flatten(1)), (t + "q txt", lambda ws: ws.QKV[:T, :dim], tq[0].flatten(1)),
            (t + "k txt", lambda ws: ws.QKV[:T, dim:2 * dim], tk[0].flatten(1)), (t + "v txt", lambda ws: ws.QKV[:T, 2 * dim:], tv[0].flatten(1))]))
        plan.append(("qkv_prepare", [(t + "q norm+rope", lambda ws: ws.Q[0], _heads_from_bshd(qr)),
                                     (t + "k norm+rope", lambda ws: ws.K[0], _heads_from_bshd(kr))]))
        plan.append(("attention_prepared", [(t + "attention", lambda ws: ws.ATT, o[0])]))
        plan.append(("gemm_grouped", [(t + "x + g*attn img", lambda ws: ws.X[T:], xa[0]), (t + "x + g*attn txt", lambda ws: ws.X[:T], ta[0])]))
        plan.append(("ln_modulate", [(t + "ln2 img", lambda ws: ws.XN[T:], in2[0]), (t + "ln2 txt", lambda ws: ws.XN[:T], tn2[0])]))
        plan.append(("gemm_grouped", [(t + "gelu(ff up) img", lambda ws: ws.FFH[T:], iffh[0]), (t + "gelu(ff up) txt", lambda ws: ws.FFH[:T], tffh[0])]))
        plan.append(("gemm_grouped", [(t + "x + g*ff img", lambda ws: ws.X[T:], xf[0]), (t + "x + g*ff txt", lambda ws: ws.X[:T], tf[0])]))
    no, po = c.take(2)
    plan.append(("ln_modulate", [("norm_out", lambda ws: ws.XN[T:], no[0])]))
    plan.append(("gemm", []))
    assert c.done(), "oracle trace longer than the plan"
    return plan, po


def wan_plan(points, cfg, affine_norm2: bool = True):
    """apex_studio_amd.wan._forward_one against oracle.wan."""
    dim = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    c = Cursor(points)
    plan: List[Stage] = []
    x0, ctxh, ctx = c.take(3)
    plan.append(("gemm", [("patch_embedding", lambda ws: ws.X, x0[0])]))
    plan.append(("gemm", [("gelu(text linear_1)", lambda ws: ws.CTXH, ctxh[0])]))
    plan.append(("gemm", [("text linear_2", lambda ws: ws.CTX, ctx[0])]))
    for b in range(cfg["num_layers"]):
        (n1, q, qn, k, kn, v, qr, kr, o, x1, n2, cq, cqn, ck, ckn, cv, cqr, ckr, co, x2, n3, h, x3) = c.take(23)
        t = f"b{b} "
        plan.append(("ln_modulate", [(t + "ln1", lambda ws: ws.XN, n1[0])]))
        plan.append(("gemm", [(t + "q", lambda ws: ws.QKV[:, :dim], q[0]), (t + "k", lambda ws: ws.QKV[:, dim:2 * dim], k[0]),
                              (t + "v", lambda ws: ws.QKV[:, 2 * dim:], v[0])]))
        plan.append(("ln_modulate", [(t + "rmsnorm q", lambda ws: ws.QKV[:, :dim], qn[0])]))
        plan.append(("ln_modulate", [(t + "rmsnorm k", lambda ws: ws.QKV[:, dim:2 * dim], kn[0])]))
        plan.append(("qkv_prepare", [(t + "q rope", lambda ws: ws.Q[0], qr[0]), (t + "k rope", lambda ws: ws.K[0], kr[0])]))
        plan.append(("attention_prepared", [(t + "self attention", lambda ws: ws.ATT, o[0])]))
        plan.append(("gemm", [(t + "x + g*attn1", lambda ws: ws.X, x1[0])]))
        if affine_norm2:
            plan.append(("ln_modulate", [(t + "norm2", lambda ws: ws.XN, n2[0])]))
        plan.append(("gemm", [(t + "cross q", lambda ws: ws.QKV[:, :dim], cq[0])]))
        plan.append(("ln_modulate", [(t + "rmsnorm cross q", lambda ws: ws.QKV[:, :dim], cqn[0])]))
        plan.append(("gemm", [(t + "cross k", lambda ws: ws.KV2[:, :dim], ck[0]), (t + "cross v", lambda ws: ws.KV2[:, dim:], cv[0])]))
        plan.append(("ln_modulate", [(t + "rmsnorm cross k", lambda ws: ws.KV2[:, :dim], ckn[0])]))
        plan.append(("qkv_prepare", [(t + "cross q heads", lambda ws: ws.Q[0], cqr[0])]))
        plan.append(("qkv_prepare", [(t + "cross k heads", lambda ws: ws.K2[0], ckr[0])]))
        plan.append(("attention_prepared", [(t + "cross attention", lambda ws: ws.ATT, co[0])]))
        plan.append(("gemm", [(t + "x + attn2", lambda ws: ws.X, x2[0])]))
        plan.append(("ln_modulate", [(t + "ln3", lambda ws: ws.XN, n3[0])]))
        plan.append(("gemm", [(t + "gelu(ffn up)", lambda ws: ws.FFH, h[0])]))
        plan.append(("gemm", [(t + "x + g*ffn", lambda ws: ws.X, x3[0])]))
    no, po = c.take(2)
    plan.append(("ln_modulate", [("norm_out", lambda ws: ws.XN, no[0])]))
    plan.append(("gemm", []))
    assert c.done(), "oracle trace longer than the plan"
    return plan, po


# ---- generic teacher forcing: match every tensor an op wrote against the oracle's unconsumed storage points ----------------
SLOT_HEADS = (2, 4, 8, 12, 16, 20, 24, 28, 32, 40)
GEN_OPS = ("gemm", "gemm_grouped", "ln_modulate", "qkv_prepare", "attention_prepared", "attention", "add_rowvec",
           "gather_rows", "attention_bias", "mul", "rope_half_")


def _op_outputs(name, args, kwargs, ret):
    """The tensors an op wrote, as views [rows, heads, width] that alias the op's storage."""
    if name == "gemm_grouped":
        outs = list(kwargs.get("out_list", args[3] if len(args) > 3 else ()))
    elif name == "qkv_prepare":
        outs = [args[4].permute(1, 0, 2), args[5].permute(1, 0, 2)]          # Q / K [H, S, 128] -> [S, H, 128]
    elif name == "attention_prepared":
        outs = [args[3][0]]                                                  # [1, S, H, 128]
    elif name == "attention":
        outs = [ret[0].permute(1, 0, 2)]                                     # [1, H, S, D] view of [1, S, H, D]
    elif name == "rope_half_":
        outs = [args[0]]                                                     # in place over the q | k columns of a fused buffer
    else:
        outs = [ret]
    return [o if o.dim() == 3 else o.unsqueeze(1) for o in outs]


def _point_rows(p: torch.Tensor, joint=None) -> torch.Tensor:
    """An oracle storage point as rows [r, c]; a joint-stream point ([latent | condition] rows in the oracle) is put into
    the HIP buffers' [condition | latent] order."""
    rows = p.reshape(-1, p.shape[-1]) if p.dim() <= 3 else p.reshape(p.shape[0] * p.shape[1], -1)
    if joint is not None and rows.shape[0] == joint[0] + joint[1]:
        rows = torch.cat([rows[joint[0]:], rows[:joint[0]]], dim=0)
    return rows


def run_forced_generic(ops_mod, points: Sequence[torch.Tensor], call: Callable[[], torch.Tensor], joint=None,
                       force: bool = True, accept: float = 2e-2, heads_first=None):
    """Teacher forcing without a hand-written plan.  Every tensor an op writes is searched for the oracle's unconsumed storage
    points: a point [r, c] may sit at the top or the bottom rows of the written tensor [R, C] (the two streams of a joint
    buffer) and at any column offset that is a multiple of c (q | k | v of a fused projection).  A right match is ~1e-5
    away, a wrong one ~1; the best candidate under `accept` claims the block, is compared, and (force) overwritten.
    Returns (output, report, indices of the points no op produced)."""
    report = []
    rows = [_point_rows(p, joint).to("cuda") for p in points]
    # a heads-first 3-D point [H, S, D] (per-sample attention operands) is stored by the HIP path as [S, H * D]
    for i, p in enumerate(points):
        if p.dim() == 3 and heads_first is not None and tuple(p.shape[1:]) == tuple(heads_first):
            rows[i] = p.permute(1, 0, 2).reshape(p.shape[1], -1).to("cuda")
    used = [False] * len(rows)
    partial = {}                 # point index -> row blocks already matched (a point written by several per-sample calls)
    orig = {n: getattr(ops_mod, n) for n in GEN_OPS}

    def claim(name, view):
        R, Hh, D = view.shape
        C = Hh * D
        flat = view.reshape(R, C).float()
        covered = []
        # the written tensor may also be ONE sample's rows of a batched point (per-sample attention calls): same width, the
        # point's row count a multiple of R
        for idx, ref in enumerate(rows):
            if used[idx] or ref.shape[1] != C or ref.shape[0] <= R or ref.shape[0] % R:
                continue
            done = partial.get(idx, set())
            for blk in range(ref.shape[0] // R):
                if blk in done:
                    continue
                want = ref[blk * R:(blk + 1) * R]
                rel = float((flat - want).norm() / (want.norm() + 1e-30))
                if rel < accept:
                    done.add(blk)
                    partial[idx] = done
                    report.append((len(report), name, f"point {idx} {tuple(points[idx].shape)} rows {blk * R}+{R}", rel,
                                   int((flat != want).sum()), want.numel()))
                    if force:
                        view.copy_(want.to(view.dtype).view(R, Hh, D))
                    if len(done) == ref.shape[0] // R:
                        used[idx] = True
                    return
        while True:
            best = None
            for idx, ref in enumerate(rows):
                if used[idx] or idx in partial or ref.shape[0] > R or ref.shape[1] > C:
                    continue
                r, c = ref.shape
                want = ref.to(flat.device)
                step = c if C % c == 0 else 64          # q | k | v blocks of unequal width (GQA): any 64-column boundary
                for r0 in {0, R - r}:
                    for c0 in range(0, C - c + 1, step):
                        if any(r0 < b and a < r0 + r and c0 < d and cc < c0 + c for a, b, cc, d in covered):
                            continue
                        g = flat[r0:r0 + r, c0:c0 + c]
                        rel = float((g - want).norm() / (want.norm() + 1e-30))
                        if rel < accept and (best is None or rel < best[0]):
                            best = (rel, idx, r0, c0, int((g != want).sum()), want, None)
                # heads narrower than the kernels' 128-wide slots (Qwen2.5-VL vision: 80): the HIP buffers keep every head in a
                # 128-column slot (zeros behind it); the point [r, nh * dv] sits in a block of nh slots
                for nh in SLOT_HEADS:
                    dv = c // nh
                    if c % nh or dv >= 128 or dv % 8 or nh * 128 > C:
                        continue
                    bw = nh * 128
                    for r0 in {0, R - r}:
                        for c0 in range(0, C - bw + 1, bw):
                            if any(r0 < b and a < r0 + r and c0 < d and cc < c0 + bw for a, b, cc, d in covered):
                                continue
                            g = flat[r0:r0 + r, c0:c0 + bw].reshape(r, nh, 128)[:, :, :dv].reshape(r, c)
                            rel = float((g - want).norm() / (want.norm() + 1e-30))
                            if rel < accept and (best is None or rel < best[0]):
                                best = (rel, idx, r0, c0, int((g != want).sum()), want, (nh, dv))
            if best is None:
                break
            rel, idx, r0, c0, nd, want, slot = best
            r, c = want.shape
            used[idx] = True
            covered.append((r0, r0 + r, c0, c0 + (c if slot is None else slot[0] * 128)))
            report.append((len(report), name, f"point {idx} {tuple(points[idx].shape)} @ rows {r0}+{r} cols {c0}+{c}"
                           + ("" if slot is None else f" in {slot[0]} x 128 slots"), rel, nd, want.numel()))
            if force:
                if slot is not None:
                    nh, dv = slot
                    tgt = view.reshape(R, C)[r0:r0 + r, c0:c0 + nh * 128].unflatten(1, (nh, 128))
                    assert tgt.data_ptr() == view.reshape(R, C)[r0:r0 + r, c0:].data_ptr(), "slot view must alias the op's storage"
                    tgt[:, :, :dv].copy_(want.to(view.dtype).view(r, nh, dv))
                elif Hh == 1:
                    view[r0:r0 + r, 0, c0:c0 + c].copy_(want.to(view.dtype))
                else:
                    assert c == C, "a point inside a heads-layout tensor must span every head"
                    view[r0:r0 + r].copy_(want.to(view.dtype).view(r, Hh, D))
                flat = view.reshape(R, C).float()

    def wrap(name):
        def f(*a, **k):
            ret = orig[name](*a, **k)
            torch.cuda.synchronize()
            for v in _op_outputs(name, a, k, ret):
                claim(name, v)
            return ret
        return f

    for n in GEN_OPS:
        setattr(ops_mod, n, wrap(n))
    try:
        out = call()
        torch.cuda.synchronize()
    finally:
        for n in GEN_OPS:
            setattr(ops_mod, n, orig[n])
    return out, report, [i for i, u in enumerate(used) if not u]
