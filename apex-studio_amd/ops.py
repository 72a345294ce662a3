"""torch.Tensor-level wrappers over the C-ABI (include/apexmi.h).

PyTorch supplies device memory and the current HIP stream; all arithmetic happens in
libapex_mi355.so.  Every wrapper refuses CPU tensors: the product path has no fallback.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib as _l


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise _l.ApexMIError(f"{name}: expected a ROCm device tensor, got {t.device} (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise _l.ApexMIError(f"{name}: expected {dtype}, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- activation storage type -----------------------------------------------------------------------------------------
# Production stores activations as bf16.  A float32 activation tensor selects the f32-STORAGE VERIFICATION MODE of the
# library (include/apexmi.h, last section): the same kernels instantiated with float storage, the MFMA
# kernels fed the exact three-way bf16 split of the activations.  Weights, gains and biases are bf16 in both modes.
_ACT = (torch.bfloat16, torch.float32)


def _req_act(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _l.ApexMIError(f"{name}: expected a ROCm device tensor, got {t.device} (no CPU fallback)")
    if t.dtype not in _ACT:
        raise _l.ApexMIError(f"{name}: activations are stored as bfloat16 (or float32 in the verification mode), got {t.dtype}")


def _fn(name: str, t: torch.Tensor):
    """The entry point `name` for t's storage type (`name_f32` for float activations)."""
    return getattr(_l.load(), name + ("_f32" if t.dtype == torch.float32 else ""))


def split3(a: torch.Tensor) -> torch.Tensor:
    """[M, K] float (row stride free) -> [M, 3K] bf16 = [hi | mid | lo], hi + mid + lo == a exactly."""
    _req(a, torch.float32, "split3.a")
    assert a.dim() == 2 and a.stride(1) == 1 and a.shape[1] % 8 == 0
    M, K = a.shape
    out = torch.empty((M, 3 * K), dtype=torch.bfloat16, device=a.device)
    _l.check(_l.load().apexmi_split_bf16x3(a.data_ptr(), a.stride(0), M, K, out.data_ptr(), out.stride(0), _stream()),
             "split_bf16x3")
    return out


_w3_cache: dict = {}
_w3_bytes = 0
_W3_CAP = 16 << 30


def clear_verification_caches() -> None:
    """Drop the repeated-weight copies the f32-storage verification mode keeps (they hold their source weights alive)."""
    global _w3_bytes
    _w3_cache.clear()
    _w3c_cache.clear()
    _w3_bytes = 0


# ---- verification through the SHIPPED kernels ------------------------------------------------------------
# The f32-storage mode replaces three kernel families by float-capable stand-ins (generic attention, the 128x128 convolution,
# the two-pass q/k/v preparation).  With this switch on, float activations still flow everywhere else, but those three points run
# the kernels production launches — the flash attention kernels, the slab / conv-shaped convolution tiles (with their fused norm
# epilogue), the fused QKV epilogue — on the bf16 ROUNDING of their float operands, and their bf16 results are widened back to
# float: the chain error then contains exactly those kernels' own storage roundings and nothing else of the bf16 path.
_shipped_verify = False


def verify_through_shipped_kernels(on: bool = True) -> None:
    global _shipped_verify
    _shipped_verify = bool(on)


def shipped_verification() -> bool:
    return _shipped_verify


def tensor_version(t: torch.Tensor):
    """`t._version`, or None for an inference tensor (created under `torch.inference_mode()`, as everything inside the
    reference's `UniversalEngine.run` is — `R/src/engine/registry.py:196`): those do not track a version counter and
    reading it raises, so callers must not cache on them across calls."""
    return None if t.is_inference() else t._version


def _w3(w: torch.Tensor) -> torch.Tensor:
    """[N, K] bf16 weight -> [N, 3K] = [w | w | w] (layout only), the partner of split3(); cached per weight version.
    An entry keeps a reference to its source view, so the address in the key cannot be recycled while the entry lives."""
    global _w3_bytes
    ver = tensor_version(w)
    if ver is None:                 # inference tensor: no version counter to key on — rebuild, never cache
        return w.repeat(1, 3).contiguous()
    key = (w.data_ptr(), ver, tuple(w.shape), w.stride(0))
    hit = _w3_cache.get(key)
    if hit is None:
        if _w3_bytes > _W3_CAP:
            clear_verification_caches()
        hit = (w, w.repeat(1, 3).contiguous())
        _w3_cache[key] = hit
        _w3_bytes += hit[1].numel() * 2
    return hit[1]


# ---- fp8-scaled weights RESIDENT in HBM (reference FPScaledLinear, R/src/quantize/scaled_layer.py:390-552) --------------
class Fp8Weight:
    """A Linear weight kept as the checkpoint stores it — float8 (e4m3fn / e5m2) [N, K] + `scale_weight` (one value, or one per
    row) — and dequantised PER CALL into a per-stream bf16 scratch right before the GEMM that reads it, as the reference's
    `FPScaledLinear.forward` does (`_scale_and_cast_weight`, scaled_layer.py:496-549: `weight.to(bf16) * scale.to(bf16)`, then a
    bf16 matmul).  Same dequantisation kernel as the load-time path (`apexmi_dequant_fp8_scaled`, every code point pinned by
    tests/golden/fp_scaled.pt), so a forward is bit-identical to dequantise-at-load; the model holds half the weight bytes."""

    def __init__(self, q: torch.Tensor, scale: torch.Tensor):
        if q.dtype not in (torch.float8_e4m3fn, torch.float8_e5m2) or q.dim() != 2:
            raise TypeError(f"Fp8Weight: expected a 2-D float8 tensor, got {q.dtype} {tuple(q.shape)}")
        self.q = q.contiguous()
        s = scale.to(q.device).to(torch.bfloat16).reshape(-1).contiguous()
        if s.numel() not in (1, q.shape[0]):
            raise ValueError(f"Fp8Weight: scale has {s.numel()} values for {q.shape[0]} rows")
        self.scale = s
        self.shape = q.shape
        self.device = q.device
        # which model Linears the rows belong to: [(module path, first row, rows)] (set by the model that adopts the record)
        self.parts: List[Tuple[str, int, int]] = []
        # RUN-TIME LoRA on a weight whose bf16 form does not exist (set_lora): lora_A [Rp, K] = the active adapters' down factors
        # stacked along the rank (zero rows up to a multiple of 64), lora_B [N, Rp] = their up factors x scale, placed on the rows
        # of their part (block-diagonal for a fused projection)
        self.lora_A: Optional[torch.Tensor] = None
        self.lora_B: Optional[torch.Tensor] = None

    @classmethod
    def cat(cls, parts: Sequence["Fp8Weight"]) -> "Fp8Weight":
        """Rows of several weights stacked (a fused q | k | v projection): per-tensor scales become per-row vectors."""
        if len({p.q.dtype for p in parts}) != 1 or len({p.shape[1] for p in parts}) != 1:
            raise ValueError("Fp8Weight.cat: parts must share the fp8 format and the input width")
        q = torch.cat([p.q.view(torch.uint8) for p in parts], dim=0).view(parts[0].q.dtype)
        s = torch.cat([p.scale if p.scale.numel() == p.shape[0] else p.scale.expand(p.shape[0]) for p in parts])
        out = cls(q, s)
        r0 = 0
        for p in parts:
            out.parts += [(m, r0 + a, n) for m, a, n in p.parts]
            r0 += p.shape[0]
        return out

    @torch.no_grad()
    def set_lora(self, adapters: Sequence[Tuple[str, torch.Tensor, torch.Tensor, float]]) -> int:
        """`adapters`: (module path, A [r, K], B [rows of that part, r], scale) of every ACTIVE adapter touching this record.
        The reference keeps fp8-scaled base weights and runs `base(x) + scale * B(A(x))` per Linear at run time (PEFT layers
        around FPScaledLinear, R/src/lora/manager.py:454-606); gemm() below does the same as ONE extra skinny GEMM and a K
        extended by the padded rank (see _gemm_fp8_lora).  Returns the padded rank (0 = no adapter left)."""
        if not adapters:
            self.lora_A = self.lora_B = None
            return 0
        N, K = self.shape
        row_of = {m: (a, n) for m, a, n in self.parts}
        R = sum(int(a.shape[0]) for _, a, _, _ in adapters)
        Rp = (R + 63) // 64 * 64
        A = torch.zeros((Rp, K), dtype=torch.float32, device=self.device)
        B = torch.zeros((N, Rp), dtype=torch.float32, device=self.device)
        c = 0
        for m, a, b, sc in adapters:
            if m not in row_of:
                raise KeyError(f"Fp8Weight.set_lora: '{m}' is not a part of this record ({[p[0] for p in self.parts]})")
            r0, n = row_of[m]
            r = int(a.shape[0])
            if tuple(a.shape) != (r, K) or tuple(b.shape) != (n, r):
                raise ValueError(f"LoRA factors {tuple(b.shape)} x {tuple(a.shape)} do not fit the {n} x {K} weight of '{m}'")
            A[c:c + r] = a.to(self.device, torch.float32)
            B[r0:r0 + n, c:c + r] = b.to(self.device, torch.float32) * float(sc)
            c += r
        self.lora_A, self.lora_B = A.to(torch.bfloat16), B.to(torch.bfloat16)
        return Rp

    def nbytes(self) -> int:
        return self.q.numel() + 2 * self.scale.numel()

    def dequant(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return dequant_fp8_scaled(self.q, self.scale, out=out)


_fp8_scratch: dict = {}


def _fp8_of(w):
    return w if isinstance(w, Fp8Weight) else getattr(w, "_fp8", None)


def _scratch(dev, n: int) -> torch.Tensor:
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    buf = _fp8_scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=torch.bfloat16, device=dev)
        _fp8_scratch[key] = buf
    return buf


def _gemm_fp8_lora(a, f8: "Fp8Weight", bias, out, epilogue, gate, residual, lora_buf):
    """epi(a W^T + (a A^T) B'^T + bias) for a resident-fp8 weight W with run-time LoRA factors, as TWO launches:
         t = a A^T                      [M, Rp] bf16 — the reference's `lora_A(x)` output, rounded as it is there — written into
                                        the columns right behind `a` in its padded buffer (`lora_buf`: same rows, >= K + Rp wide);
         out = epi([a | t] [W | B']^T)  one GEMM over K + Rp: W dequantised into the per-stream scratch with row stride K + Rp
                                        (the kernel of the load-time path), B' = scale x B copied behind it.
    The adapter term is accumulated in f32 with the base product and rounded ONCE, under any epilogue (GELU included) — at least as
    accurate as the reference's bf16 `result + lora_B(lora_A(x)) * scaling`.  Cost: Rp / K more K-tiles (1.25 % at rank 64 on
    d = 5120) + the skinny GEMM; no pass over the [M, N] output."""
    M, K = a.shape
    N = f8.shape[0]
    Rp = f8.lora_A.shape[0]
    if a.dtype != torch.bfloat16:
        raise NotImplementedError("run-time LoRA on resident-fp8 weights exists for bf16 activation storage only")
    if lora_buf is None or lora_buf.data_ptr() != a.data_ptr() or lora_buf.stride(0) != a.stride(0) or lora_buf.shape[1] < K + Rp \
            or lora_buf.shape[0] != M:
        raise _l.ApexMIError(f"gemm: the weight carries run-time LoRA factors (rank {Rp} padded) and needs `lora_buf` = the activation "
                             f"buffer of `a` with >= {K + Rp} columns (got {None if lora_buf is None else tuple(lora_buf.shape)})")
    gemm(a, f8.lora_A, None, out=lora_buf[:, K:K + Rp])
    w2 = _scratch(f8.device, N * (K + Rp))[:N * (K + Rp)].view(N, K + Rp)
    f8.dequant(out=w2[:, :K])
    w2[:, K:].copy_(f8.lora_B)
    return gemm(lora_buf[:, :K + Rp], w2, bias, out=out, epilogue=epilogue, gate=gate, residual=residual)


def _bf16_weight(w, float_acts: bool = False):
    """`w` as the bf16 [N, K] operand of a GEMM launched next on the current stream: the tensor itself, or — for an Fp8Weight / a
    parameter carrying one (`param._fp8`, weights.load_checkpoint_into(keep_fp8=True)) — its dequantisation into the stream's
    scratch.  Stream order makes the reuse safe: the next dequantisation is queued behind the GEMM that read the last one."""
    f8 = w if isinstance(w, Fp8Weight) else getattr(w, "_fp8", None)
    if f8 is None:
        return w
    if f8.lora_A is not None:
        raise _l.ApexMIError("this resident-fp8 weight carries run-time LoRA factors: only gemm(..., lora_buf=) applies them")
    if float_acts:           # verification mode caches on (data_ptr, version): never hand it a recycled buffer
        return f8.dequant()
    n = f8.shape[0] * f8.shape[1]
    return f8.dequant(out=_scratch(f8.device, n)[:n].view(f8.shape[0], f8.shape[1]))


_EPI = {"bias": _l.EPI_BIAS, "gelu": _l.EPI_BIAS_GELU, "gate_res": _l.EPI_BIAS_GATE_RES,
        "gelu_erf": _l.EPI_BIAS_GELU_ERF, "silu": _l.EPI_BIAS_SILU, "quick_gelu": _l.EPI_BIAS_QUICK_GELU}


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, epilogue: str = "bias",
         gate: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         lora_buf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  2-D operands, last dim contiguous; a / out / residual bf16, or float32
    in the f32-storage verification mode (w and bias stay bf16).  `lora_buf`: see _gemm_fp8_lora (ignored unless `w` is a
    resident-fp8 weight with run-time LoRA factors)."""
    _req_act(a, "gemm.a")
    f8 = _fp8_of(w)
    if f8 is not None and f8.lora_A is not None:
        return _gemm_fp8_lora(a, f8, bias, out, epilogue, gate, residual, lora_buf)
    w = _bf16_weight(w, a.dtype == torch.float32)
    _req(w, torch.bfloat16, "gemm.w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    act = a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=act, device=a.device)
    else:
        _req(out, act, "gemm.out")
        assert out.shape == (M, N) and out.stride(1) == 1
    if bias is not None:
        _req(bias, torch.bfloat16, "gemm.bias")
        assert bias.is_contiguous() and bias.numel() == N
    ldr = 0
    if epilogue == "gate_res":
        _req(gate, torch.float32, "gemm.gate")
        _req(residual, act, "gemm.residual")
        assert gate.is_contiguous() and gate.numel() == N
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    epi = _EPI[epilogue]
    if act == torch.float32:     # exact bf16 split of the activations against [w | w | w]: same kernel, float epilogue
        a, w, K, epi = split3(a), _w3(w), 3 * K, epi | _l.EPI_F32_IO
    rc = _l.load().apexmi_gemm_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _ptr(bias),
                                    out.data_ptr(), out.stride(0), M, N, K, epi,
                                    _ptr(gate), _ptr(residual), ldr, _stream())
    _l.check(rc, "gemm_bf16")
    return out


def gemm_grouped(a_list, w_list, bias_list, out_list, epilogue="bias", gate_list=None,
                 residual_list=None) -> None:
    """Several gemm() problems sharing K in one launch (img/txt streams; QKV + MLP-up of a single
    block).  `epilogue` is one name or a list with one name per problem."""
    import ctypes as C
    n = len(a_list)
    K = w_list[0].shape[1]
    epis = [epilogue] * n if isinstance(epilogue, str) else list(epilogue)
    act = a_list[0].dtype
    # resident fp8 weights: the problems of one launch need their dequantised operands side by side — carved out of ONE per-stream
    # scratch (stream order makes the reuse safe, see _bf16_weight); a weight with run-time LoRA factors has no grouped form
    f8s = [_fp8_of(w) for w in w_list]
    if any(f is not None for f in f8s):
        if any(f is not None and f.lora_A is not None for f in f8s):
            raise _l.ApexMIError("gemm_grouped: a resident-fp8 weight carries run-time LoRA factors: only gemm(..., lora_buf=) applies them")
        if act == torch.float32:     # verification mode caches on (data_ptr, version): never hand it a recycled buffer
            w_list = [w if f is None else f.dequant() for w, f in zip(w_list, f8s)]
        else:
            sizes = [0 if f is None else (f.shape[0] * f.shape[1] + 7) // 8 * 8 for f in f8s]
            dev = next(f for f in f8s if f is not None).device
            buf, off, ws = _scratch(dev, sum(sizes)), 0, []
            for w, f, n in zip(w_list, f8s, sizes):
                ws.append(w if f is None else f.dequant(out=buf[off:off + f.shape[0] * f.shape[1]].view(f.shape[0], f.shape[1])))
                off += n
            w_list = ws
    for a, w, o in zip(a_list, w_list, out_list):
        _req_act(a, "gemm_grouped.a")
        _req(a, act, "gemm_grouped.a")
        _req(w, torch.bfloat16, "gemm_grouped.w")
        _req(o, act, "gemm_grouped.out")
        assert a.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and o.stride(1) == 1
        assert w.shape[1] == K and a.shape[1] == K and o.shape == (a.shape[0], w.shape[0])
    VP = C.c_void_p * n
    I64 = C.c_int64 * n
    INT = C.c_int * n
    none = [None] * n
    bias_list = bias_list or none
    gate_list = gate_list or none
    residual_list = residual_list or none
    for e, g, r, o, w in zip(epis, gate_list, residual_list, out_list, w_list):
        if e == "gate_res":
            _req(g, torch.float32, "gemm_grouped.gate")
            _req(r, act, "gemm_grouped.residual")
            assert g.is_contiguous() and g.numel() == w.shape[0] and r.shape == o.shape and r.stride(1) == 1
    flag = 0
    if act == torch.float32:     # f32-storage verification mode (see gemm)
        done: dict = {}
        sp = []
        for a in a_list:         # problems that read the same activations share one split
            k_ = (a.data_ptr(), tuple(a.shape), a.stride(0))
            if k_ not in done:
                done[k_] = split3(a)
            sp.append(done[k_])
        a_list, w_list, K, flag = sp, [_w3(w) for w in w_list], 3 * K, _l.EPI_F32_IO
    rc = _l.load().apexmi_gemm_bf16_grouped(
        n, VP(*[a.data_ptr() for a in a_list]), I64(*[a.stride(0) for a in a_list]),
        VP(*[w.data_ptr() for w in w_list]), I64(*[w.stride(0) for w in w_list]),
        VP(*[_ptr(b) for b in bias_list]), VP(*[o.data_ptr() for o in out_list]),
        I64(*[o.stride(0) for o in out_list]), INT(*[a.shape[0] for a in a_list]),
        INT(*[w.shape[0] for w in w_list]), K, INT(*[_EPI[e] | flag for e in epis]),
        VP(*[_ptr(g) for g in gate_list]), VP(*[_ptr(r) for r in residual_list]),
        I64(*[(r.stride(0) if r is not None else 0) for r in residual_list]), _stream())
    _l.check(rc, "gemm_bf16_grouped")


def qkv_fusable(a_list, w_list, row0, H: int) -> bool:
    """Would gemm_grouped_qkv() take this launch?  (bf16 storage, an even head count, the 256x256 16x16x32 tiling.)"""
    if H % 2 or any(a.dtype != torch.bfloat16 and not (_shipped_verify and a.dtype == torch.float32) for a in a_list):
        return False
    return bool(_l.load().apexmi_gemm_qkv_fusable(sum(a.shape[0] for a in a_list), max(w.shape[0] for w in w_list),
                                                  w_list[0].shape[1]))


def gemm_grouped_qkv(a_list, w_list, bias_list, out_list, epilogue, is_qkv, norm_q, norm_k, row0, H: int, eps: float,
                     rope: torch.Tensor, qo: torch.Tensor, ko: torch.Tensor, vt: torch.Tensor) -> None:
    """gemm_grouped() with qkv_prepare() fused into the epilogue of the problems flagged in `is_qkv` (fused QKV projections,
    N = 3 H 128): q / k leave normalised per head, rotated and laid out [H, S_out, 128], v leaves transposed [H, 128, Skp];
    bit-identical to gemm_grouped + qkv_prepare.  out_list[i] of a fused problem is ignored (may be None).  bf16 storage only;
    raises (apexmi error) when the launch would not run on the 256x256 16x16x32 tiling — callers keep the two-pass path then."""
    import ctypes as C
    n = len(a_list)
    K = w_list[0].shape[1]
    epis = [epilogue] * n if isinstance(epilogue, str) else list(epilogue)
    if _shipped_verify and any(a.dtype == torch.float32 for a in a_list):
        # verification through the shipped kernels: the fused problems take the bf16 rounding of their float operand (the value
        # production stores there) and leave bf16 q / k / v^T; problems that ride along (the single block's MLP-up) keep the
        # float path, as their own launch
        none_ = [None] * n
        bl, nq_, nk_ = bias_list or none_, norm_q or none_, norm_k or none_
        keep = [i for i in range(n) if is_qkv[i]]
        for i in range(n):
            if not is_qkv[i]:
                gemm(a_list[i], w_list[i], bl[i], out=out_list[i], epilogue=epis[i])
        sel = lambda lst: [lst[i] for i in keep]          # noqa: E731
        return gemm_grouped_qkv([to_bf16(a.contiguous()) if a.dtype == torch.float32 else a for a in sel(a_list)], sel(w_list),
                                sel(bl), [None] * len(keep), sel(epis), [1] * len(keep), sel(nq_), sel(nk_), sel(list(row0)), H, eps,
                                rope, qo, ko, vt)
    for a, w, o, f in zip(a_list, w_list, out_list, is_qkv):
        _req(a, torch.bfloat16, "gemm_grouped_qkv.a")
        _req(w, torch.bfloat16, "gemm_grouped_qkv.w")
        assert a.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K and a.shape[1] == K
        if not f:
            _req(o, torch.bfloat16, "gemm_grouped_qkv.out")
            assert o.stride(1) == 1 and o.shape == (a.shape[0], w.shape[0])
    for t_ in (qo, ko, vt):
        _req(t_, torch.bfloat16, "gemm_grouped_qkv output")
        assert t_.is_contiguous()
    _req(rope, torch.float32, "gemm_grouped_qkv.rope")
    S_out = qo.shape[1]
    assert qo.shape == ko.shape == (H, S_out, 128) and vt.shape[:2] == (H, 128) and rope.is_contiguous()
    assert rope.shape == (2, S_out, 128)
    VP = C.c_void_p * n
    I64 = C.c_int64 * n
    INT = C.c_int * n
    none = [None] * n
    bias_list = bias_list or none
    pairs = rope_pairs(rope)
    rc = _l.load().apexmi_gemm_bf16_grouped_qkv_pairs(
        n, VP(*[a.data_ptr() for a in a_list]), I64(*[a.stride(0) for a in a_list]),
        VP(*[w.data_ptr() for w in w_list]), I64(*[w.stride(0) for w in w_list]), VP(*[_ptr(b) for b in bias_list]),
        VP(*[_ptr(o) if not f else None for o, f in zip(out_list, is_qkv)]),
        I64(*[(o.stride(0) if (o is not None and not f) else 0) for o, f in zip(out_list, is_qkv)]),
        INT(*[a.shape[0] for a in a_list]), INT(*[w.shape[0] for w in w_list]), K, INT(*[_EPI[e] for e in epis]),
        INT(*[1 if f else 0 for f in is_qkv]), VP(*[_ptr(t_) for t_ in (norm_q or none)]), VP(*[_ptr(t_) for t_ in (norm_k or none)]),
        INT(*[int(r) for r in row0]), int(H), float(eps), rope.data_ptr(), _ptr(pairs), qo.data_ptr(), ko.data_ptr(), vt.data_ptr(),
        S_out, vt.shape[2], _stream())
    _l.check(rc, "gemm_bf16_grouped_qkv")


rope_pairs_enabled = True      # A/B switch (tests, tools): False = the fused epilogue reads the full [2, S, 128] table


def rope_pairs(rope: torch.Tensor, trusted: bool = False) -> Optional[torch.Tensor]:
    """The compact copy [2, S, D / 2] of a rotary table whose entries come in equal pairs (apexmi_rope_table_axes writes every
    cos / sin twice), made ONCE per table (cached on the tensor; tables are cached per geometry by the models) — the fused q/k/v
    epilogue prefetches its rows from it.  None for a table that is not pair-duplicated (checked on the device, one host read per
    table).  `trusted`: the caller made the table with `rope_table_axes` (pairs by construction): no check, no host read — the
    models pass it, so a table rebuilt per call (ids that are inference tensors) costs one more small launch, not a sync."""
    if not rope_pairs_enabled:
        return None
    key = (rope.data_ptr(), tensor_version(rope))
    hit = getattr(rope, "_apex_pairs", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    _req(rope, torch.float32, "rope_pairs.rope")
    assert rope.dim() == 3 and rope.shape[0] == 2 and rope.is_contiguous() and rope.shape[2] % 2 == 0
    S, D = rope.shape[1], rope.shape[2]
    out = torch.empty((2, S, D // 2), dtype=torch.float32, device=rope.device)
    bad = torch.zeros((), dtype=torch.int32, device=rope.device)
    _l.check(_l.load().apexmi_rope_pairs(rope.data_ptr(), S, D, out.data_ptr(), bad.data_ptr(), _stream()), "rope_pairs")
    res = out if (trusted or int(bad.item()) == 0) else None
    try:
        rope._apex_pairs = (key, res)
    except Exception:
        pass
    return res


def gemv(w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, pre_silu: bool = False, post: Optional[str] = None,
         accum: bool = False) -> torch.Tensor:
    """y[M,N] (f32) = post(pre(x[M,K] f32) @ w[N,K]^T (bf16) + bias)."""
    _req(w, torch.bfloat16, "gemv.w")
    _req(x, torch.float32, "gemv.x")
    assert w.dim() == 2 and w.stride(1) == 1 and x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        assert not accum
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        _req(out, torch.float32, "gemv.out")
        assert out.shape == (M, N) and out.stride(1) == 1
    if bias is not None:
        _req(bias, torch.bfloat16, "gemv.bias")
    flags = 0
    if pre_silu:
        flags |= _l.GEMV_PRE_SILU
    if post == "silu":
        flags |= _l.GEMV_POST_SILU
    elif post == "gelu":
        flags |= _l.GEMV_POST_GELU
    elif post is not None:
        raise ValueError(post)
    if accum:
        flags |= _l.GEMV_ACCUM
    rc = _l.load().apexmi_gemv(w.data_ptr(), w.stride(0), _ptr(bias), x.data_ptr(), x.stride(0),
                               out.data_ptr(), out.stride(0), M, N, K, flags, _stream())
    _l.check(rc, "gemv")
    return out


def ln_modulate(x: torch.Tensor, scale: Optional[torch.Tensor] = None,
                shift: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
                eps: float = 1e-6, rms: bool = False, split: int = 0,
                scale2: Optional[torch.Tensor] = None, shift2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = LayerNorm(x) [*gamma + beta] * (1 + scale) + shift, or RMSNorm(x) * gamma.
    Rows [0, split) use (scale2, shift2) instead (text rows of a joint buffer)."""
    _req_act(x, "ln_modulate.x")
    assert x.dim() == 2 and x.stride(1) == 1
    M, Cc = x.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=x.dtype, device=x.device)
    else:
        _req(out, x.dtype, "ln_modulate.out")
        assert out.shape == (M, Cc) and out.stride(1) == 1
    for t, nm in ((scale, "scale"), (shift, "shift"), (scale2, "scale2"), (shift2, "shift2")):
        if t is not None:
            _req(t, torch.float32, "ln_modulate." + nm)
            assert t.is_contiguous() and t.numel() == Cc
    for t, nm in ((gamma, "gamma"), (beta, "beta")):
        if t is not None:
            _req(t, torch.bfloat16, "ln_modulate." + nm)
            assert t.is_contiguous() and t.numel() == Cc
    rc = _fn("apexmi_ln_modulate2", x)(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, Cc,
                                        _ptr(scale), _ptr(shift), _ptr(gamma), _ptr(beta), float(eps),
                                        1 if rms else 0, int(split), _ptr(scale2), _ptr(shift2), _stream())
    _l.check(rc, "ln_modulate")
    return out


def qkv_prepare(q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor], H: int,
                qo: torch.Tensor, ko: torch.Tensor, vt: Optional[torch.Tensor],
                wq: Optional[torch.Tensor] = None, wk: Optional[torch.Tensor] = None,
                wq2: Optional[torch.Tensor] = None, wk2: Optional[torch.Tensor] = None,
                split: int = 0, eps: float = 1e-6, rope: Optional[torch.Tensor] = None,
                rope_mode: int = _l.ROPE_NONE, row0: int = 0) -> None:
    """q,k,v: [S, H*128] row views sharing one row stride; qo,ko: [H, S_out, 128]; vt: [H,128,Skp]."""
    _req_act(q, "qkv_prepare.q")
    for t_ in (k, v, qo, ko, vt):
        if t_ is not None:
            _req(t_, q.dtype, "qkv_prepare operand")
    S = q.shape[0]
    D = q.shape[1] // H
    ld = q.stride(0)
    assert (k is None or k.stride(0) == ld) and (v is None or v.stride(0) == ld)
    assert qo.is_contiguous() and qo.shape[0] == H
    assert k is None or (ko.is_contiguous() and qo.shape == ko.shape)
    S_out = qo.shape[1]
    Skp = 0
    if vt is not None:
        assert vt.is_contiguous() and vt.shape[0] == H and vt.shape[1] == D
        Skp = vt.shape[2]
    if rope is not None:
        _req(rope, torch.float32, "qkv_prepare.rope")
        assert rope.is_contiguous()
    rc = _fn("apexmi_qkv_prepare", q)(q.data_ptr(), _ptr(k), _ptr(v), ld, S, H, D, split,
                                       _ptr(wq), _ptr(wk), _ptr(wq2), _ptr(wk2), float(eps),
                                       _ptr(rope), rope_mode, qo.data_ptr(), _ptr(ko), _ptr(vt),
                                       S_out, Skp, row0, _stream())
    _l.check(rc, "qkv_prepare")


def qk_rms_rope_rows(q: torch.Tensor, k: Optional[torch.Tensor], v: Optional[torch.Tensor], H: int, qo: torch.Tensor,
                     ko: Optional[torch.Tensor], vt: Optional[torch.Tensor], wq: Optional[torch.Tensor] = None,
                     wk: Optional[torch.Tensor] = None, eps: float = 1e-6, rope: Optional[torch.Tensor] = None,
                     rope_mode: int = _l.ROPE_NONE, row0: int = 0) -> None:
    """Wan's q / k preparation in one pass: RMSNorm over all H * 128 channels of every q (and k) row (weights wq / wk of that
    width), rounded to the storage type where the in-place norm of the reference writes it, RoPE, [H, S_out, 128] layout; v
    (optional) transposed to [H, 128, Skp].  Bit-identical to ln_modulate(rms) on q, on k, then qkv_prepare."""
    _req_act(q, "qk_rms_rope_rows.q")
    for t_ in (k, v, qo, ko, vt):
        if t_ is not None:
            _req(t_, q.dtype, "qk_rms_rope_rows operand")
    S, ld = q.shape[0], q.stride(0)
    assert q.shape[1] == H * 128 and (k is None or k.stride(0) == ld) and (v is None or v.stride(0) == ld)
    assert qo.is_contiguous() and qo.shape[0] == H and (k is None) == (ko is None) and (v is None) == (vt is None)
    assert ko is None or (ko.is_contiguous() and ko.shape == qo.shape)
    Skp = 0
    if vt is not None:
        assert vt.is_contiguous() and vt.shape[0] == H and vt.shape[1] == 128
        Skp = vt.shape[2]
    for w_ in (wq, wk):
        if w_ is not None:
            _req(w_, torch.bfloat16, "qk_rms_rope_rows weight")
            assert w_.is_contiguous() and w_.numel() == H * 128
    if rope is not None:
        _req(rope, torch.float32, "qk_rms_rope_rows.rope")
        assert rope.is_contiguous()
    rc = _fn("apexmi_qk_rms_rope_rows", q)(q.data_ptr(), _ptr(k), _ptr(v), ld, S, H, _ptr(wq), _ptr(wk), float(eps), _ptr(rope),
                                           rope_mode, qo.data_ptr(), _ptr(ko), _ptr(vt), qo.shape[1], Skp, row0, _stream())
    _l.check(rc, "qk_rms_rope_rows")


def v_transpose(v: torch.Tensor, vt: torch.Tensor) -> None:
    """v: [S, H, 128] view (any row/head stride) -> vt [H, 128, Skp] zero padded."""
    _req(v, torch.bfloat16, "v_transpose.v")
    S, H, D = v.shape
    assert v.stride(2) == 1 and vt.is_contiguous()
    rc = _l.load().apexmi_v_transpose(v.data_ptr(), v.stride(1), v.stride(0), S, H, D, vt.data_ptr(),
                                      vt.shape[2], 0, _stream())
    _l.check(rc, "v_transpose")


def attention_prepared(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor,
                       Sk: int, scale: Optional[float] = None) -> torch.Tensor:
    """q [B,H,Sq,128], k [B,H,Sk,128], vt [B,H,128,Skp] packed; out [B,Sq,H,128] (strided ok)."""
    _req_act(q, "attention.q")
    B, H, Sq, D = q.shape
    assert D == 128 and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    assert out.shape == (B, Sq, H, D) and out.stride(3) == 1
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _l.load()
    if _shipped_verify and (q.dtype == torch.float32 or out.dtype == torch.float32):
        # verification through the shipped flash kernels: bf16 roundings of q / k / v^T in (or the fused epilogue's bf16
        # outputs as they are), the kernel's bf16 result widened into the float buffer
        qb, kb, vb = (to_bf16(t_) if t_.dtype == torch.float32 else t_ for t_ in (q, k, vt))
        ob = torch.empty((B, Sq, H, D), dtype=torch.bfloat16, device=q.device)
        attention_prepared(qb, kb, vb, ob, Sk, scale)
        out.copy_(ob)
        return out
    if q.dtype == torch.float32:     # f32-storage verification mode: f32 arithmetic, no bf16 probabilities
        for t_ in (k, vt, out):
            _req(t_, torch.float32, "attention operand")
        rc = lib.apexmi_attn_fwd_prepared_f32(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, H, Sq, Sk,
                                              vt.shape[3], _l.i64x3((out.stride(0), out.stride(1), out.stride(2))),
                                              float(scale), _stream())
        _l.check(rc, "attn_fwd_prepared_f32")
        return out
    need = lib.apexmi_attn_prepared_workspace_bytes(B, H, Sq, Sk)     # scratch of the tail split, usually 0
    ws = None
    if need:
        key = ("prep", q.device.index, torch.cuda.current_stream().cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q.device)
            _ws_cache[key] = ws
    rc = lib.apexmi_attn_fwd_prepared_ws(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, H, Sq, Sk,
                                         vt.shape[3], _l.i64x3((out.stride(0), out.stride(1), out.stride(2))),
                                         float(scale), _ptr(ws), need, _stream())
    _l.check(rc, "attn_fwd_prepared")
    return out


_DT = {torch.bfloat16: _l.BF16, torch.float16: _l.F16, torch.float32: _l.F32}
_ws_cache: dict = {}


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
              softmax_scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q k^T scale) v for q:[B,H,Sq,D], k,v:[B,H,Sk,D] (permuted views welcome).
    Returns a [B,H,Sq,D] view of a [B,Sq,H,D] buffer, i.e. `.permute(0,2,1,3)` by the caller
    (flux/base/attention.py:95) is a no-copy view and `.flatten(2,3)` stays a view."""
    _req(q, None, "attention.q")
    if q.dtype not in _DT or k.dtype != q.dtype or v.dtype != q.dtype:
        raise _l.ApexMIError(f"attention: unsupported dtypes {q.dtype}/{k.dtype}/{v.dtype}")
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    if q.stride(3) != 1:
        q = q.contiguous()
    if k.stride(3) != 1:
        k = k.contiguous()
    if v.stride(3) != 1:
        v = v.contiguous()
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(D)
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    lib = _l.load()
    need = lib.apexmi_attn_workspace_bytes(B, H, Sq, Sk, D, _DT[q.dtype])
    ws = None
    if need:
        key = (q.device.index, torch.cuda.current_stream().cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q.device)
            _ws_cache[key] = ws
    rc = lib.apexmi_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Sq, Sk, D,
                             _l.i64x3((q.stride(0), q.stride(1), q.stride(2))),
                             _l.i64x3((k.stride(0), k.stride(1), k.stride(2))),
                             _l.i64x3((v.stride(0), v.stride(1), v.stride(2))),
                             _l.i64x3((out.stride(0), out.stride(1), out.stride(2))),
                             float(softmax_scale), _DT[q.dtype], _ptr(ws), need, _stream())
    _l.check(rc, "attn_fwd")
    return out.permute(0, 2, 1, 3)


def attention_framecausal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tokens_per_frame: int,
                          softmax_scale: Optional[float] = None) -> torch.Tensor:
    """Frame-causal attention of the HunyuanVideo-1.5 VAE mid block: q, k, v [B,H,S,D] bf16 (D a multiple of 128 up to
    1024, any S), token i attends the keys of frames <= its own.  Same return convention as `attention`."""
    _req(q, torch.bfloat16, "attention_framecausal.q")
    B, H, S, D = q.shape
    q, k, v = (t if t.stride(3) == 1 else t.contiguous() for t in (q, k, v))
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(D)
    out = torch.empty((B, S, H, D), dtype=q.dtype, device=q.device)
    lib = _l.load()
    need = lib.apexmi_attn_framecausal_workspace_bytes(S, D)
    key = (q.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < max(need, 1):
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=q.device)
        _ws_cache[key] = ws
    rc = lib.apexmi_attn_fwd_framecausal(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, S, D,
                                         int(tokens_per_frame),
                                         _l.i64x3((q.stride(0), q.stride(1), q.stride(2))),
                                         _l.i64x3((k.stride(0), k.stride(1), k.stride(2))),
                                         _l.i64x3((v.stride(0), v.stride(1), v.stride(2))),
                                         _l.i64x3((out.stride(0), out.stride(1), out.stride(2))),
                                         float(softmax_scale), ws.data_ptr(), need, _stream())
    _l.check(rc, "attn_fwd_framecausal")
    return out.permute(0, 2, 1, 3)


def attention_bias(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, softmax_scale: float,
                   bias: Optional[torch.Tensor] = None, keep: Optional[torch.Tensor] = None,
                   causal: bool = False, out: Optional[torch.Tensor] = None, kv_heads: Optional[int] = None,
                   seg: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Text-encoder self-attention over packed projections: q [Sq, H*D], k, v [Sk, Hkv*D] bf16 (row-strided views of a
    fused QKV buffer welcome), bias f32 [H, Sq, Sk] (T5 relative-position bias), keep uint8 [Sk] (0 = padded key), seg
    int32 [S] (block-diagonal attention: segment id per token).  Returns [Sq, H*D]."""
    _req_act(q, "attention_bias.q")
    _req(k, q.dtype, "attention_bias.k")
    _req(v, q.dtype, "attention_bias.v")
    assert q.dim() == 2 and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    Sq, inner = q.shape
    Sk = k.shape[0]
    D = inner // heads
    if out is None:
        out = torch.empty((Sq, inner), dtype=q.dtype, device=q.device)
    if bias is not None:
        _req(bias, torch.float32, "attention_bias.bias")
        assert bias.is_contiguous() and bias.shape == (heads, Sq, Sk)
    if keep is not None:
        _req(keep, torch.uint8, "attention_bias.keep")
        assert keep.is_contiguous() and keep.numel() == Sk
    if seg is not None:
        _req(seg, torch.int32, "attention_bias.seg")
        assert seg.is_contiguous() and seg.numel() == Sq == Sk
    hkv = heads if kv_heads is None else int(kv_heads)
    assert k.shape[1] == hkv * D and v.shape[1] == hkv * D
    lib = _l.load()
    if q.dtype == torch.float32:           # f32-storage verification mode: one f32 pass per (row, head), no bf16 probabilities
        _req(out, torch.float32, "attention_bias.out")
        rc = lib.apexmi_attn_fwd_bias_f32(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                          out.data_ptr(), out.stride(0), heads, hkv, Sq, Sk, D, float(softmax_scale), _ptr(bias),
                                          _ptr(keep), _ptr(seg), 1 if causal else 0, _stream())
        _l.check(rc, "attn_fwd_bias_f32")
        return out
    need = lib.apexmi_attn_bias_workspace_bytes(heads, Sq, Sk, D)
    key = (q.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        _ws_cache[key] = ws
    rc = lib.apexmi_attn_fwd_bias(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                  out.data_ptr(), out.stride(0), heads, hkv, Sq, Sk, D, float(softmax_scale), _ptr(bias),
                                  _ptr(keep), _ptr(seg), 1 if causal else 0, ws.data_ptr(), need, _stream())
    _l.check(rc, "attn_fwd_bias")
    return out


def rope_half_(x: torch.Tensor, heads: int, head_stride: int, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """In place: every head of x [rows, >= heads * head_stride] (bf16, row-strided view welcome) rotates its first D =
    cos.shape[1] columns, x <- x cos + rotate_half(x) sin with f32 tables [rows, D]."""
    _req_act(x, "rope_half.x")
    _req(cos, torch.float32, "rope_half.cos")
    _req(sin, torch.float32, "rope_half.sin")
    assert x.dim() == 2 and x.stride(1) == 1 and cos.shape == sin.shape == (x.shape[0], cos.shape[1])
    assert cos.is_contiguous() and sin.is_contiguous() and x.shape[1] >= heads * head_stride
    _l.check(_fn("apexmi_rope_half", x)(x.data_ptr(), x.stride(0), x.shape[0], heads, head_stride, cos.shape[1],
                                        cos.data_ptr(), sin.data_ptr(), _stream()), "rope_half")
    return x


def relpos_bias(weight: torch.Tensor, bucket: torch.Tensor, Sq: int, Sk: int) -> torch.Tensor:
    """weight bf16 [num_buckets, H], bucket int32 [Sq + Sk - 1] (device) -> f32 [H, Sq, Sk]."""
    _req(weight, torch.bfloat16, "relpos_bias.weight")
    _req(bucket, torch.int32, "relpos_bias.bucket")
    assert weight.is_contiguous() and bucket.is_contiguous() and bucket.numel() == Sq + Sk - 1
    nb, H = weight.shape
    out = torch.empty((H, Sq, Sk), dtype=torch.float32, device=weight.device)
    _l.check(_l.load().apexmi_relpos_bias(weight.data_ptr(), nb, H, bucket.data_ptr(), Sq, Sk, out.data_ptr(), _stream()),
             "relpos_bias")
    return out


def gather_rows(table: torch.Tensor, ids: torch.Tensor, pos: Optional[torch.Tensor] = None,
                out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """table[ids] (+ pos[row % len(pos)]): bf16 [vocab, C], ids int64 [rows] on the device -> [rows, C] (`out_dtype` float32: the
    f32-storage verification mode, the position sum unrounded)."""
    _req(table, torch.bfloat16, "gather_rows.table")
    _req(ids, torch.int64, "gather_rows.ids")
    assert table.dim() == 2 and table.stride(1) == 1 and ids.dim() == 1 and ids.is_contiguous()
    rows, Cc = ids.numel(), table.shape[1]
    out = torch.empty((rows, Cc), dtype=out_dtype, device=table.device)
    period = 1
    if pos is not None:
        _req(pos, torch.bfloat16, "gather_rows.pos")
        assert pos.dim() == 2 and pos.stride(1) == 1 and pos.shape[1] == Cc
        period = pos.shape[0]
    if out_dtype == torch.float32:
        rc = _l.load().apexmi_gather_rows_f32(table.data_ptr(), table.stride(0), table.shape[0], ids.data_ptr(), _ptr(pos),
                                              pos.stride(0) if pos is not None else 0, period, out.data_ptr(), out.stride(0),
                                              rows, Cc, _stream())
        _l.check(rc, "gather_rows_f32")
        return out
    rc = _l.load().apexmi_gather_rows_bf16(table.data_ptr(), table.stride(0), table.shape[0], ids.data_ptr(), _ptr(pos),
                                           pos.stride(0) if pos is not None else 0, period, out.data_ptr(), out.stride(0),
                                           rows, Cc, _stream())
    _l.check(rc, "gather_rows_bf16")
    return out


def frames_to_u8(video: torch.Tensor) -> torch.Tensor:
    """bf16 video [C, T, H, W] (any strides) in [-1, 1] -> uint8 frames [T, H, W, C]."""
    _req_act(video, "frames_to_u8.video")
    assert video.dim() == 4
    Cc, T, H, W = video.shape
    out = torch.empty((T, H, W, Cc), dtype=torch.uint8, device=video.device)
    rc = _fn("apexmi_frames_to_u8", video)(video.data_ptr(), video.stride(0), video.stride(1), video.stride(2), video.stride(3),
                                       Cc, T, H, W, out.data_ptr(), _stream())
    _l.check(rc, "frames_to_u8")
    return out


def mul(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a * b for contiguous bf16 tensors of equal shape (numel a multiple of 8); float tensors in the f32-storage mode."""
    _req_act(a, "mul.a")
    _req(b, a.dtype, "mul.b")
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    if a.dtype == torch.float32:
        _l.check(_l.load().apexmi_mul_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "mul_f32")
        return out
    _l.check(_l.load().apexmi_mul_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "mul_bf16")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a + b for contiguous tensors of equal shape (bf16; float in the f32-storage mode; numel a multiple of 8)."""
    _req_act(a, "add.a")
    _req(b, a.dtype, "add.b")
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    fn = getattr(_l.load(), "apexmi_add_f32" if a.dtype == torch.float32 else "apexmi_add_bf16")
    _l.check(fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def group_mean(x: torch.Tensor, out_channels: int) -> torch.Tensor:
    """Mean over consecutive channel groups of a channels-last tensor: [..., C * gs] -> [..., C] (f32 sum; bf16, or float in the
    f32-storage mode)."""
    _req_act(x, "group_mean.x")
    assert x.is_contiguous() and x.shape[-1] % out_channels == 0
    gs = x.shape[-1] // out_channels
    out = torch.empty(*x.shape[:-1], out_channels, dtype=x.dtype, device=x.device)
    P = x.numel() // x.shape[-1]
    fn = getattr(_l.load(), "apexmi_group_mean_f32" if x.dtype == torch.float32 else "apexmi_group_mean_bf16")
    _l.check(fn(x.data_ptr(), out.data_ptr(), P, out_channels, gs, _stream()), "group_mean")
    return out


_freq_tables: dict = {}


def _timestep_freqs(dim: int, shift: float, device) -> torch.Tensor:
    """The layer constant exp(-ln(10000) * i / (dim/2 - shift)), i < dim/2, built ONCE per (dim, shift, device) on the
    host with the f32 operation sequence of diffusers' `get_timestep_embedding` (in-tree copy: reference
    transformer/qwenimage/base/model.py:46-97), so the kernel's sin / cos arguments equal the reference's bit for bit."""
    key = (dim, float(shift), str(device))
    f = _freq_tables.get(key)
    if f is None:
        half = dim // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
        f = torch.exp(exponent / (half - shift)).to(device)
        _freq_tables[key] = f
    return f


def timestep_embedding(t: torch.Tensor, dim: int, scale: float = 1.0, flip_sin_to_cos: bool = True,
                       downscale_freq_shift: float = 0.0) -> torch.Tensor:
    _req(t, torch.float32, "timestep_embedding.t")
    t = t.contiguous()
    M = t.numel()
    out = torch.empty((M, dim), dtype=torch.float32, device=t.device)
    freqs = _timestep_freqs(dim, downscale_freq_shift, t.device)
    rc = _l.load().apexmi_timestep_embedding(t.data_ptr(), out.data_ptr(), M, dim, float(scale),
                                             1 if flip_sin_to_cos else 0, float(downscale_freq_shift),
                                             freqs.data_ptr(), _stream())
    _l.check(rc, "timestep_embedding")
    return out


def rope_table_axes(ids: torch.Tensor, axes_dim, theta: float = 10000.0) -> torch.Tensor:
    """ids f32 [S, n_axes] -> f32 [2, S, sum(axes_dim)] (cos plane, sin plane)."""
    import ctypes as C
    _req(ids, torch.float32, "rope_table_axes.ids")
    ids = ids.contiguous()
    S, n = ids.shape
    D = int(sum(axes_dim))
    out = torch.empty((2, S, D), dtype=torch.float32, device=ids.device)
    arr = (C.c_int * n)(*[int(a) for a in axes_dim])
    rc = _l.load().apexmi_rope_table_axes(ids.data_ptr(), S, n, arr, float(theta), out.data_ptr(), _stream())
    _l.check(rc, "rope_table_axes")
    return out


def add_bcast(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[l, :] = a[l, :] + b  (f32; a [L, n] contiguous, b [n])."""
    _req(a, torch.float32, "add_bcast.a")
    _req(b, torch.float32, "add_bcast.b")
    assert a.is_contiguous() and b.is_contiguous() and a.shape[-1] == b.numel()
    if out is None:
        out = torch.empty_like(a)
    rows = a.numel() // b.numel()
    _l.check(_l.load().apexmi_add_bcast_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, b.numel(),
                                            _stream()), "add_bcast_f32")
    return out


def add_rowvec(x: torch.Tensor, v: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r, :] = x[r, :] + v  (bf16 [rows, cols] + bf16 [cols], f32 add; x / out float in the f32-storage mode)."""
    _req_act(x, "add_rowvec.x")
    _req(v, torch.bfloat16, "add_rowvec.v")
    assert x.dim() == 2 and x.stride(1) == 1 and v.is_contiguous() and v.numel() == x.shape[1]
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    else:
        _req(out, x.dtype, "add_rowvec.out")
        assert out.shape == x.shape and out.stride(1) == 1
    fn = getattr(_l.load(), "apexmi_add_rowvec_f32" if x.dtype == torch.float32 else "apexmi_add_rowvec_bf16")
    _l.check(fn(x.data_ptr(), x.stride(0), v.data_ptr(), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], _stream()),
             "add_rowvec")
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32, "to_bf16.x")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().apexmi_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()))
    return out


def dequant_fp8_scaled(w: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[rows, cols] (bf16) = w (float8_e4m3fn | float8_e5m2, [rows, cols]).to(bf16) * scale.to(bf16); `scale` is
    a scalar or one value per row.  `out` may be a row-range view of a packed weight matrix."""
    fmt = {torch.float8_e4m3fn: 0, torch.float8_e5m2: 1}.get(w.dtype)
    if fmt is None:
        raise TypeError(f"dequant_fp8_scaled: weight dtype {w.dtype} is not an fp8 format")
    if not w.is_cuda:
        raise RuntimeError("dequant_fp8_scaled: tensors must be on the GPU (there is no CPU path)")
    w2 = w.reshape(w.shape[0], -1).contiguous()
    rows, cols = w2.shape
    s = scale.to(w.device).to(torch.bfloat16).reshape(-1).contiguous()
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.bfloat16, device=w.device)
    else:
        _req(out, torch.bfloat16, "dequant_fp8_scaled.out")
        assert out.shape == (rows, cols) and out.stride(1) == 1
    _l.check(_l.load().apexmi_dequant_fp8_scaled(w2.data_ptr(), fmt, s.data_ptr(), s.numel(), rows, cols,
                                                 out.data_ptr(), out.stride(0), _stream()), "dequant_fp8_scaled")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.bfloat16, "to_f32.x")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _l.check(_l.load().apexmi_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()))
    return out


def euler_step(sample: torch.Tensor, model_output: torch.Tensor, dt: float,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """prev = sample + dt * model_output in f32, stored in sample's dtype (bf16 or f32)."""
    _req(model_output, torch.bfloat16, "euler_step.model_output")
    _req(sample, None, "euler_step.sample")
    assert sample.is_contiguous() and model_output.is_contiguous()
    assert sample.numel() == model_output.numel()
    if out is None:
        out = torch.empty_like(sample)
    rc = _l.load().apexmi_euler_step(sample.data_ptr(), model_output.data_ptr(), out.data_ptr(),
                                     sample.numel(), float(dt), _DT[sample.dtype], _stream())
    _l.check(rc, "euler_step")
    return out


# ---- channels-last VAE ops ---------------------------------------------------------------------
_zero_lines: dict = {}


def _zeros16(device) -> torch.Tensor:
    z = _zero_lines.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.bfloat16, device=device)
        _zero_lines[device] = z
    return z


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, (kT,) kH, kW] -> [Cout4, Kpad] with k = tap*Cin + ci, zero padded (K to 64, Cout to 4)."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin = w.shape[:2]
    k = w.shape[2] * w.shape[3] * w.shape[4] * cin
    kpad = (k + 63) // 64 * 64
    if w.shape[2] > 1:     # room to address the last temporal slice on its own (single-frame / independent-frame launches)
        spatial = w.shape[3] * w.shape[4] * cin
        kpad = (max(kpad, (w.shape[2] - 1) * spatial + (spatial + 63) // 64 * 64) + 63) // 64 * 64
    cout4 = (cout + 3) // 4 * 4
    out = torch.zeros(cout4, kpad, dtype=w.dtype, device=w.device)
    out[:cout, :k] = w.permute(0, 2, 3, 4, 1).reshape(cout, k)
    return out


_w3c_cache: dict = {}


def _conv_w3(w_packed: torch.Tensor, ksize, cin: int) -> torch.Tensor:
    """Packed conv weight [Cout4, Kpad] (k = tap * cin + ci) -> the packing of the same weight with every tap's channel run
    repeated three times (k = tap * 3 cin + j cin + ci): the partner of the [hi | mid | lo] channel split."""
    ver = tensor_version(w_packed)
    key = (w_packed.data_ptr(), ver, tuple(w_packed.shape), tuple(ksize), cin)
    hit = None if ver is None else _w3c_cache.get(key)   # (source, repeated): the source reference pins the address in the key
    t = None if hit is None else hit[1]
    if t is None:
        kT, kH, kW = (int(v) for v in ksize)
        taps = kT * kH * kW
        cout4 = w_packed.shape[0]
        k3 = taps * 3 * cin
        kpad = (k3 + 63) // 64 * 64
        if kT > 1:
            spatial = kH * kW * 3 * cin
            kpad = (max(kpad, (kT - 1) * spatial + (spatial + 63) // 64 * 64) + 63) // 64 * 64
        t = torch.zeros(cout4, kpad, dtype=w_packed.dtype, device=w_packed.device)
        t[:, :k3] = w_packed[:, :taps * cin].reshape(cout4, taps, 1, cin).expand(cout4, taps, 3, cin).reshape(cout4, k3)
        if len(_w3c_cache) > 512:
            _w3c_cache.clear()
        if ver is not None:
            _w3c_cache[key] = (w_packed, t)
    return t


def _conv_f32(x, w_packed, bias, ksize, out_shape, residual=None, out=None, replicate=False, independent_frames=False,
              upsample2x=False, stride=(1, 1), pad=(-1, -1), out_hw=(0, 0), tstride=(1, 0, 0), slope=None):
    """The f32-storage verification form of every conv3d_cl* wrapper below: float activations in, float out / residual."""
    _req(x, torch.float32, "conv3d_cl.x")
    T, H, W, cin = x.shape
    x3 = split3(x.view(T * H * W, cin)).view(T, H, W, 3 * cin)
    w3 = _conv_w3(w_packed, ksize, cin)
    cout = w_packed.shape[0]
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and tuple(out.shape) == tuple(out_shape) and out.dtype == torch.float32
    if residual is not None:
        _req(residual, torch.float32, "conv3d_cl.residual")
        assert residual.shape == out.shape and residual.is_contiguous()
    flags = (1 if replicate else 0) | (2 if independent_frames else 0) | (4 if upsample2x else 0)
    rc = _l.load().apexmi_conv3d_cl_f32(x3.data_ptr(), w3.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                        _zeros16(x.device).data_ptr(), T, H, W, 3 * cin, cout, w3.shape[1], int(ksize[0]),
                                        int(ksize[1]), int(ksize[2]), flags, int(stride[0]), int(stride[1]), int(pad[0]),
                                        int(pad[1]), int(out_hw[0]), int(out_hw[1]), int(tstride[0]), int(tstride[1]),
                                        int(tstride[2]), 0 if slope is None else 1, 0.0 if slope is None else float(slope),
                                        _stream())
    _l.check(rc, "conv3d_cl_f32")
    return out


def conv3d_cl(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ksize,
              residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              replicate: bool = False, independent_frames: bool = False, upsample2x: bool = False,
              clip_frames: int = 0) -> torch.Tensor:
    """x [T,H,W,Cin] bf16 contiguous -> [T,H,W,Cout4]; causal in time, "same" padding in space: zeros, or (replicate)
    clamped coordinates as HunyuanVideo15CausalConv3d pads.  upsample2x: the convolution reads x through a nearest 2x
    spatial upsample (output [T,2H,2W,Cout4]) without materialising it.  clip_frames: the T frames are T / clip_frames
    independent clips stacked along T (the tiles of a tiled decode in one launch, `apexmi_conv3d_cl_clips`)."""
    _req_act(x, "conv3d_cl.x")
    _req(w_packed, torch.bfloat16, "conv3d_cl.w")
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, cin = x.shape
    cout, kpad = w_packed.shape
    Ho, Wo = (2 * H, 2 * W) if upsample2x else (H, W)
    if x.dtype == torch.float32 and _shipped_verify and not (clip_frames and clip_frames != T):
        # verification through the shipped convolution tiles (slab / conv-shaped / 128x128, whichever production picks for this
        # shape): bf16 rounding of the float input, the kernel's bf16 result widened; the residual is added in float
        y = to_f32(conv3d_cl(to_bf16(x), w_packed, bias, ksize, replicate=replicate, independent_frames=independent_frames,
                             upsample2x=upsample2x))
        if residual is not None:
            y = add(y, residual)
        if out is not None:
            out.copy_(y)
            return out
        return y
    if x.dtype == torch.float32:
        assert not (replicate and independent_frames) and not (replicate and upsample2x)
        if clip_frames and clip_frames != T:
            raise _l.ApexMIError("conv3d_cl: stacked clips are not part of the f32-storage verification mode")
        return _conv_f32(x, w_packed, bias, ksize, (T, Ho, Wo, cout), residual=residual, out=out, replicate=replicate,
                         independent_frames=independent_frames, upsample2x=upsample2x)
    if out is None:
        out = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.shape == (T, Ho, Wo, cout)
    if bias is not None:
        assert bias.numel() == cout and bias.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    assert not (replicate and independent_frames) and not (replicate and upsample2x)
    if upsample2x:
        rc = _l.load().apexmi_conv3d_cl_up2(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                            _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, int(ksize[0]),
                                            int(ksize[1]), int(ksize[2]), 1 if independent_frames else 0, _stream())
        _l.check(rc, "conv3d_cl_up2")
        return out
    if clip_frames and clip_frames != T:
        assert not independent_frames and T % int(clip_frames) == 0
        rc = _l.load().apexmi_conv3d_cl_clips(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                              _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, int(ksize[0]),
                                              int(ksize[1]), int(ksize[2]), 1 if replicate else 0, int(clip_frames), _stream())
        _l.check(rc, "conv3d_cl_clips")
        return out
    fn = (_l.load().apexmi_conv3d_cl_replicate if replicate else
          _l.load().apexmi_conv3d_cl_frames if independent_frames else _l.load().apexmi_conv3d_cl)
    rc = fn(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
            _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, int(ksize[0]), int(ksize[1]), int(ksize[2]),
            _stream())
    _l.check(rc, "conv3d_cl")
    return out


def conv3d_cl_act(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ksize,
                  residual: Optional[torch.Tensor] = None, slope: Optional[float] = None, upsample2x: bool = False,
                  independent_frames: bool = False) -> torch.Tensor:
    """act(conv(x) + bias (+ residual)) with act = leaky ReLU(slope) (slope None: no activation), zero padding, causal in
    time; upsample2x / independent_frames as conv3d_cl.  The TAEHV blocks (reference vae/tae/model.py:20-45)."""
    _req_act(x, "conv3d_cl_act.x")
    _req(w_packed, torch.bfloat16, "conv3d_cl_act.w")
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, cin = x.shape
    cout, kpad = w_packed.shape
    Ho, Wo = (2 * H, 2 * W) if upsample2x else (H, W)
    if x.dtype == torch.float32:
        return _conv_f32(x, w_packed, bias, ksize, (T, Ho, Wo, cout), residual=residual, independent_frames=independent_frames,
                         upsample2x=upsample2x, slope=slope)
    out = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        assert bias.numel() == cout and bias.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    rc = _l.load().apexmi_conv3d_cl_act(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                        _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, int(ksize[0]), int(ksize[1]),
                                        int(ksize[2]), 1 if independent_frames else 0, 1 if upsample2x else 0,
                                        0 if slope is None else 1, 0.0 if slope is None else float(slope), _stream())
    _l.check(rc, "conv3d_cl_act")
    return out


def tanh_clamp(x: torch.Tensor, inv_scale: float = 1.0) -> torch.Tensor:
    """3 tanh(x * inv_scale / 3) (TAEHV `Clamp`, reference vae/tae/model.py:24-26)."""
    _req_act(x, "tanh_clamp.x")
    assert x.is_contiguous() and (x.dtype == torch.float32 or x.numel() % 8 == 0)
    out = torch.empty_like(x)
    _l.check(_fn("apexmi_tanh_clamp", x)(x.data_ptr(), out.data_ptr(), x.numel(), float(inv_scale), _stream()), "tanh_clamp")
    return out


def pixel_shuffle_clamp(x: torch.Tensor, channels: int, r: int, trim: int = 0, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """x [T, H, W, Cs] channels-last -> [channels, T - trim, H r, W r]: clamp, F.pixel_shuffle(r), drop `trim` leading frames
    (reference vae/tae/model.py:318-333)."""
    _req_act(x, "pixel_shuffle_clamp.x")
    assert x.dim() == 4 and x.is_contiguous()
    T, H, W, cs = x.shape
    out = torch.empty((channels, T - trim, H * r, W * r), dtype=x.dtype, device=x.device)
    _l.check(_fn("apexmi_pixel_shuffle_clamp", x)(x.data_ptr(), out.data_ptr(), T, H, W, cs, channels, r, trim, float(lo),
                                                  float(hi), _stream()), "pixel_shuffle_clamp")
    return out


def conv3d_cl_norm_fusable(x: torch.Tensor, cout: int, upsample2x: bool = False) -> bool:
    T, H, W, cin = x.shape
    return bool(_l.load().apexmi_conv3d_cl_norm_fusable(T, H, W, cin, cout, 1 if upsample2x else 0))


def conv3d_cl_norm(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ksize, gamma: torch.Tensor,
                   silu: bool = True, residual: Optional[torch.Tensor] = None, want_raw: bool = True,
                   upsample2x: bool = False, independent_frames: bool = False):
    """conv3d_cl with the RMS norm (+ SiLU) of its output fused into the epilogue: returns (y or None, norm(y)).
    Shapes the fused tiles do not cover (`conv3d_cl_norm_fusable`) run as conv3d_cl + rmsnorm_cl — same values."""
    _req_act(x, "conv3d_cl_norm.x")
    _req(gamma, torch.bfloat16, "conv3d_cl_norm.gamma")
    cout, kpad = w_packed.shape
    if (x.dtype == torch.float32 and _shipped_verify and gamma.numel() == cout and conv3d_cl_norm_fusable(x, cout, upsample2x)):
        # verification through the shipped FUSED tile: conv + bias (+ bf16-rounded residual, as production holds it) + RMS norm
        # (+ SiLU) in one epilogue on the bf16 rounding of the float input; both results widened
        y, yn = conv3d_cl_norm(to_bf16(x), w_packed, bias, ksize, gamma, silu=silu,
                               residual=None if residual is None else to_bf16(residual), want_raw=want_raw,
                               upsample2x=upsample2x, independent_frames=independent_frames)
        return (None if y is None else to_f32(y)), to_f32(yn)
    # (float activations — the verification mode — otherwise take the two-launch form: same values, no fused tile)
    if x.dtype == torch.float32 or gamma.numel() != cout or not conv3d_cl_norm_fusable(x, cout, upsample2x):
        if gamma.numel() != cout:
            # the RMS norm divides by sqrt(channel count): a gamma shorter than the packed Cout (a Cout that is not a multiple
            # of 4 gets zero rows in the packed weight) would silently normalise over the padded width
            raise _l.ApexMIError(f"conv3d_cl_norm: gamma has {gamma.numel()} entries for {cout} (packed) output channels; the "
                                 f"fused / separate RMS norm needs the layer's channel count to equal the packed one")
        y = conv3d_cl(x, w_packed, bias, ksize, residual=residual, upsample2x=upsample2x, independent_frames=independent_frames)
        return (y if want_raw else None), rmsnorm_cl(y, gamma, silu=silu)
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous() and gamma.is_contiguous()
    T, H, W, cin = x.shape
    Ho, Wo = (2 * H, 2 * W) if upsample2x else (H, W)
    y = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device) if want_raw else None
    yn = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        assert residual.shape == yn.shape and residual.is_contiguous()
    rc = _l.load().apexmi_conv3d_cl_norm(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(residual), _ptr(y), yn.data_ptr(),
                                         gamma.data_ptr(), 1 if silu else 0, _zeros16(x.device).data_ptr(), T, H, W, cin, cout,
                                         kpad, int(ksize[0]), int(ksize[1]), int(ksize[2]), 1 if independent_frames else 0,
                                         1 if upsample2x else 0, _stream())
    _l.check(rc, "conv3d_cl_norm")
    return y, yn


def conv3d_cl_tstrided(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ksize, stride_t: int,
                       t_first: int, out_frames: int) -> torch.Tensor:
    """Causal conv3d evaluated only at input frames t_first, t_first + stride_t, ...: x [T,H,W,Cin] -> [out_frames,H,W,Cout4]
    (WanResample "downsample3d" time_conv, reference vae/wan/model.py:340-365)."""
    _req_act(x, "conv3d_cl_tstrided.x")
    _req(w_packed, torch.bfloat16, "conv3d_cl_tstrided.w")
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, cin = x.shape
    cout, kpad = w_packed.shape
    if x.dtype == torch.float32:
        return _conv_f32(x, w_packed, bias, ksize, (out_frames, H, W, cout), tstride=(stride_t, t_first, out_frames))
    out = torch.empty((out_frames, H, W, cout), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        assert bias.numel() == cout and bias.is_contiguous()
    rc = _l.load().apexmi_conv3d_cl_tstrided(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), None, out.data_ptr(),
                                             _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, int(ksize[0]), int(ksize[1]),
                                             int(ksize[2]), int(stride_t), int(t_first), int(out_frames), _stream())
    _l.check(rc, "conv3d_cl_tstrided")
    return out


def conv2d_cl_down2(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """ZeroPad2d((0, 1, 0, 1)) + Conv2d(3x3, stride 2) per frame: x [T, H, W, Cin] -> [T, Ho, Wo, Cout4] with
    Ho = (H - 2) // 2 + 1 (= H / 2 for even H); w_packed from pack_conv_weight of the [Cout, Cin, 3, 3] weight."""
    _req_act(x, "conv2d_cl_down2.x")
    _req(w_packed, torch.bfloat16, "conv2d_cl_down2.w")
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, cin = x.shape
    cout, kpad = w_packed.shape
    Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    if x.dtype == torch.float32:
        return _conv_f32(x, w_packed, bias, (1, 3, 3), (T, Ho, Wo, cout), stride=(2, 2), pad=(0, 0), out_hw=(Ho, Wo))
    out = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        assert bias.numel() == cout and bias.is_contiguous()
    rc = _l.load().apexmi_conv3d_cl_strided(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), None, out.data_ptr(),
                                            _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, 1, 3, 3, 2, 2, 0, 0,
                                            Ho, Wo, _stream())
    _l.check(rc, "conv3d_cl_strided")
    return out


def conv2d_cl_strided(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 2, pad: int = 1) -> torch.Tensor:
    """nn.Conv2d(3x3, stride, padding=pad) per frame: x [T, H, W, Cin] -> [T, Ho, Wo, Cout4], Ho = (H + 2 pad - 3) // stride + 1
    (the TAEHV encoder's downsampling convolutions, reference vae/tae/model.py:218, :223, :228)."""
    _req_act(x, "conv2d_cl_strided.x")
    _req(w_packed, torch.bfloat16, "conv2d_cl_strided.w")
    assert x.dim() == 4 and x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, cin = x.shape
    cout, kpad = w_packed.shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    if x.dtype == torch.float32:
        return _conv_f32(x, w_packed, bias, (1, 3, 3), (T, Ho, Wo, cout), stride=(stride, stride), pad=(pad, pad), out_hw=(Ho, Wo))
    out = torch.empty((T, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        assert bias.numel() == cout and bias.is_contiguous()
    rc = _l.load().apexmi_conv3d_cl_strided(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), None, out.data_ptr(),
                                            _zeros16(x.device).data_ptr(), T, H, W, cin, cout, kpad, 1, 3, 3, stride, stride, pad, pad,
                                            Ho, Wo, _stream())
    _l.check(rc, "conv3d_cl_strided")
    return out


def rmsnorm_cl(x: torch.Tensor, gamma: torch.Tensor, silu: bool = False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req_act(x, "rmsnorm_cl.x")
    assert x.is_contiguous() and gamma.is_contiguous() and gamma.numel() == x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    P = x.numel() // x.shape[-1]
    _l.check(_fn("apexmi_rmsnorm_cl", x)(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), P, x.shape[-1],
                                         1 if silu else 0, _stream()), "rmsnorm_cl")
    return out


def upsample2x_cl(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.bfloat16, "upsample2x_cl.x")
    assert x.dim() == 4 and x.is_contiguous()
    T, H, W, Cc = x.shape
    out = torch.empty((T, 2 * H, 2 * W, Cc), dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().apexmi_upsample2x_cl(x.data_ptr(), out.data_ptr(), T, H, W, Cc, _stream()), "upsample2x_cl")
    return out


def time_interleave_cl(x: torch.Tensor) -> torch.Tensor:
    """[T,H,W,2C] -> [2T,H,W,C]."""
    _req_act(x, "time_interleave_cl.x")
    assert x.dim() == 4 and x.is_contiguous() and x.shape[3] % 16 == 0
    T, H, W, C2 = x.shape
    out = torch.empty((2 * T, H, W, C2 // 2), dtype=x.dtype, device=x.device)
    _l.check(_fn("apexmi_time_interleave_cl", x)(x.data_ptr(), out.data_ptr(), T, H * W, C2 // 2, _stream()),
             "time_interleave_cl")
    return out


def crossfade_(a: torch.Tensor, b: torch.Tensor, dim: int) -> torch.Tensor:
    """In place on b: b[.., e, ..] = a[.., e, ..] (1 - e/E) + b[.., e, ..] e/E along `dim` (E = size of dim).
    a, b: same-shape bf16 views of [T, H, W, C] tensors (blend_v: dim=1, blend_h: dim=2)."""
    _req_act(a, "crossfade.a")
    _req(b, a.dtype, "crossfade.b")
    assert a.shape == b.shape and a.dim() == 4 and a.stride(3) == 1 and b.stride(3) == 1
    T, H, W, Cc = b.shape
    E = b.shape[dim]
    fade = _fn("apexmi_crossfade", a)
    if dim == 1:    # rows: outer = T, e = H rows, inner = W*C (needs contiguous rows in both)
        assert a.stride(2) == Cc and b.stride(2) == Cc
        rc = fade(a.data_ptr(), b.data_ptr(), T, E, W * Cc, a.stride(0), a.stride(1),
                                  b.stride(0), b.stride(1), _stream())
        _l.check(rc, "crossfade")
    else:           # columns: one call per frame, outer = H, e = W columns, inner = C
        for t in range(T):
            rc = fade(a[t].data_ptr(), b[t].data_ptr(), H, E, Cc, a.stride(1), a.stride(2),
                                      b.stride(1), b.stride(2), _stream())
            _l.check(rc, "crossfade")
    return b


_gn_ws: dict = {}


def groupnorm_cl(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-6,
                 silu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm over a channels-last image [..., C] (all leading dims are positions)."""
    _req_act(x, "groupnorm_cl.x")
    _req(gamma, torch.bfloat16, "groupnorm_cl.gamma")
    assert x.is_contiguous() and gamma.is_contiguous() and beta.is_contiguous()
    Cc = x.shape[-1]
    P = x.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    lib = _l.load()
    need = lib.apexmi_groupnorm_workspace_bytes(P, Cc)
    key = (x.device, _stream())
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    rc = _fn("apexmi_groupnorm_cl", x)(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), P, Cc, groups,
                                 float(eps), 1 if silu else 0, ws.data_ptr(), need, _stream())
    _l.check(rc, "groupnorm_cl")
    return out


# ---- independent items on side streams ----------------------------------------------------------------------------
def run_on_streams(streams: list, ns: int, n_items: int, fn, device):
    """`fn(i)` for i in range(n_items) — independent pieces of one call (the images of a batch) — side by side on `ns` HIP
    streams forked from the calling stream and joined back into it; `streams` is the caller's pool (grown on demand).  Each
    result is marked as used by the calling stream (the caching allocator keys blocks on the stream that allocated them).  The
    callee's scratch must be per stream (the models key their workspaces on the stream; the op workspaces above already do)."""
    main = torch.cuda.current_stream()
    if len(streams) < ns:
        streams += [torch.cuda.Stream(device=device) for _ in range(ns - len(streams))]
    for s_ in streams[:ns]:
        s_.wait_stream(main)
    outs = []
    for i in range(n_items):
        st = streams[i % ns]
        with torch.cuda.stream(st):
            y = fn(i)
            y.record_stream(main)
            outs.append(y)
    for s_ in streams[:ns]:
        main.wait_stream(s_)
    return outs


# ---- device guard ------------------------------------------------------------------------------------------------
# Every wrapper above enqueues on `torch.cuda.current_stream()` of the CURRENT device.  A model living on cuda:1
# while the caller's current device is cuda:0 would otherwise launch on device 0's stream with device-1 pointers.
# Each public op therefore runs with its first device operand's device current; model entry points (forward /
# decode / encode) do the same with `on_model_device`, which also covers their own stream / event objects.
def _first_device(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor) and a[0].is_cuda:
            return a[0].device
    return None


def _guarded(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = _first_device(args, kwargs)
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def on_model_device(method):
    """Decorator for model entry points: run with `self.device` as the current device."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        dev = self.device
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return method(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return method(self, *args, **kwargs)
    return wrapper


def _install_guards():
    import types
    g = globals()
    for name, obj in list(g.items()):
        if isinstance(obj, types.FunctionType) and not name.startswith("_") and obj.__module__ == __name__ \
                and name not in ("on_model_device",):
            g[name] = _guarded(obj)


_install_guards()
