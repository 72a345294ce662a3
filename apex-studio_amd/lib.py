"""ctypes binding of libapex_mi355.so (the C-ABI declared in include/apexmi.h).

There is no CPU fallback: every product entry point goes through this library, and
`load()` raises if it has not been built (run `python __graft_entry__.py` / `build.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# APEX_MI355_LIB: load another build of the library (the ablation builds of tools/gemm_ring_ablate.sh); default = the in-tree one
LIB_PATH = os.environ.get("APEX_MI355_LIB") or os.path.join(_PKG_DIR, "libapex_mi355.so")

_lib = None

c_i64p = C.POINTER(C.c_int64)
c_f32p = C.c_void_p  # device pointers travel as plain addresses
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/apexmi.h one-to-one
SIGNATURES = {
    "apexmi_version": (C.c_int, []),
    "apexmi_last_error": (C.c_char_p, []),
    "apexmi_attn_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "apexmi_attn_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  c_i64p, c_i64p, c_i64p, c_i64p, C.c_float, C.c_int, vp,
                                  C.c_size_t, vp]),
    "apexmi_attn_fwd_prepared": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, c_i64p, C.c_float, vp]),
    "apexmi_attn_prepared_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "apexmi_attn_fwd_prepared_ws": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, c_i64p, C.c_float, vp, C.c_size_t, vp]),
    "apexmi_attn_w64_fallbacks": (C.c_int, [C.POINTER(C.c_uint64)]),
    "apexmi_attn_framecausal_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "apexmi_attn_fwd_framecausal": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              c_i64p, c_i64p, c_i64p, c_i64p, C.c_float, vp, C.c_size_t, vp]),
    "apexmi_gemm_bf16": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, C.c_int, C.c_int,
                                   C.c_int, C.c_int, vp, vp, C.c_int64, vp]),
    "apexmi_gemm_bf16_grouped": (C.c_int, [C.c_int, C.POINTER(vp), c_i64p, C.POINTER(vp), c_i64p,
                                           C.POINTER(vp), C.POINTER(vp), c_i64p, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(vp),
                                           C.POINTER(vp), c_i64p, vp]),
    "apexmi_gemm_bf16_grouped_qkv": (C.c_int, [C.c_int, C.POINTER(vp), c_i64p, C.POINTER(vp), c_i64p, C.POINTER(vp),
                                               C.POINTER(vp), c_i64p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp),
                                               C.POINTER(C.c_int), C.c_int, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "apexmi_gemm_bf16_grouped_qkv_pairs": (C.c_int, [C.c_int, C.POINTER(vp), c_i64p, C.POINTER(vp), c_i64p, C.POINTER(vp),
                                                     C.POINTER(vp), c_i64p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp),
                                                     C.POINTER(C.c_int), C.c_int, C.c_float, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "apexmi_rope_pairs": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
    "apexmi_gemm_qkv_fusable": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "apexmi_gemm_uses_x288": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "apexmi_qk_rms_rope_rows": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, C.c_float, vp, C.c_int, vp, vp, vp,
                                          C.c_int, C.c_int, C.c_int, vp]),
    "apexmi_qk_rms_rope_rows_f32": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, C.c_float, vp, C.c_int, vp, vp, vp,
                                              C.c_int, C.c_int, C.c_int, vp]),
    "apexmi_gemm_bf16_batched": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "apexmi_attn_bias_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "apexmi_attn_fwd_bias": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, C.c_int, vp, C.c_size_t, vp]),
    "apexmi_rope_half": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "apexmi_rope_half_f32": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "apexmi_tune_set": (C.c_int, [C.c_char_p, C.c_int]),
    "apexmi_gemv": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int,
                              C.c_int, C.c_int, vp]),
    "apexmi_ln_modulate": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp,
                                     C.c_float, C.c_int, vp]),
    "apexmi_ln_modulate2": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp,
                                      C.c_float, C.c_int, C.c_int, vp, vp, vp]),
    "apexmi_qkv_prepare": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                     vp, vp, C.c_float, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                     C.c_int, vp]),
    "apexmi_v_transpose": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, vp, C.c_int,
                                     C.c_int, vp]),
    "apexmi_conv3d_cl": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "apexmi_conv3d_cl_replicate": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "apexmi_conv3d_cl_clips": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 11 + [vp]),
    "apexmi_conv3d_cl_strided": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 15 + [vp]),
    "apexmi_conv3d_cl_frames": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "apexmi_conv3d_cl_up2": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 10 + [vp]),
    "apexmi_conv3d_cl_norm_fusable": (C.c_int, [C.c_int] * 6),
    "apexmi_conv3d_cl_norm": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp] + [C.c_int] * 11 + [vp]),
    "apexmi_add_bf16": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "apexmi_add_f32": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "apexmi_group_mean_bf16": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "apexmi_group_mean_f32": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "apexmi_rmsnorm_cl": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "apexmi_upsample2x_cl": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "apexmi_time_interleave_cl": (C.c_int, [vp, vp, C.c_int, C.c_int64, C.c_int, vp]),
    "apexmi_conv3d_cl_act": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 12 + [C.c_float, vp]),
    "apexmi_conv3d_cl_tstrided": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 12 + [vp]),
    "apexmi_tanh_clamp": (C.c_int, [vp, vp, C.c_int64, C.c_float, vp]),
    "apexmi_pixel_shuffle_clamp": (C.c_int, [vp, vp] + [C.c_int] * 7 + [C.c_float, C.c_float, vp]),
    "apexmi_tanh_clamp_f32": (C.c_int, [vp, vp, C.c_int64, C.c_float, vp]),
    "apexmi_pixel_shuffle_clamp_f32": (C.c_int, [vp, vp] + [C.c_int] * 7 + [C.c_float, C.c_float, vp]),
    "apexmi_groupnorm_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "apexmi_groupnorm_cl": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, vp,
                                      C.c_size_t, vp]),
    "apexmi_crossfade": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, vp]),
    "apexmi_timestep_embedding": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                            vp, vp]),
    "apexmi_rope_table_axes": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_float, vp, vp]),
    "apexmi_add_bcast_f32": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, vp]),
    "apexmi_add_rowvec_bf16": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int, vp]),
    "apexmi_add_rowvec_f32": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int, vp]),
    "apexmi_frames_to_u8": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      vp, vp]),
    "apexmi_mul_bf16": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "apexmi_mul_f32": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "apexmi_gather_rows_f32": (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int, vp, C.c_int64,
                                         C.c_int64, C.c_int, vp]),
    "apexmi_attn_fwd_bias_f32": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, C.c_int, vp]),
    "apexmi_gather_rows_bf16": (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int, vp, C.c_int64,
                                          C.c_int64, C.c_int, vp]),
    "apexmi_relpos_bias": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "apexmi_cast_f32_to_bf16": (C.c_int, [vp, vp, C.c_int64, vp]),
    "apexmi_cast_bf16_to_f32": (C.c_int, [vp, vp, C.c_int64, vp]),
    "apexmi_dequant_fp8_scaled": (C.c_int, [vp, C.c_int, vp, C.c_int64, C.c_int64, C.c_int64, vp, C.c_int64, vp]),
    "apexmi_euler_step": (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, C.c_int, vp]),
    "apexmi_prof_enable": (C.c_int, [C.c_int]),
    "apexmi_prof_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "apexmi_prof_reset": (C.c_int, []),
    "apexmi_clk_enable": (C.c_int, [C.c_int]),
    "apexmi_clk_read": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}

# f32-storage verification mode (include/apexmi.h, last section): same argument lists as the bf16 entry points
for _name in ("apexmi_ln_modulate2", "apexmi_qkv_prepare", "apexmi_rmsnorm_cl", "apexmi_groupnorm_cl",
              "apexmi_time_interleave_cl", "apexmi_crossfade", "apexmi_frames_to_u8"):
    SIGNATURES[_name + "_f32"] = SIGNATURES[_name]
SIGNATURES["apexmi_attn_fwd_prepared_f32"] = SIGNATURES["apexmi_attn_fwd_prepared"]
SIGNATURES["apexmi_split_bf16x3"] = (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int64, vp])
SIGNATURES["apexmi_conv3d_cl_f32"] = (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 20 + [C.c_float, vp])

NCLASS = 6
PROF_CLASSES = ("gemm", "attention", "gemv", "ln_modulate", "qkv_prepare", "other")

BF16, F16, F32 = 0, 1, 2
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES, EPI_BIAS_F32, EPI_BIAS_GELU_ERF, EPI_BIAS_SILU, EPI_BIAS_QUICK_GELU = 0, 1, 2, 3, 4, 5, 6
EPI_F32_IO = 0x100   # C and R are float: the f32-storage verification mode
GEMV_PRE_SILU, GEMV_POST_SILU, GEMV_POST_GELU, GEMV_ACCUM = 1, 2, 4, 8
ROPE_INTERLEAVED, ROPE_COMPLEX, ROPE_NONE = 0, 1, 2


class ApexMIError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApexMIError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python __graft_entry__.py` or apex_studio_amd.build.build()). "
            "There is no CPU fallback for the product path."
        )
    # torch's bundled HIP runtime must be the one in the process before this library resolves
    # libamdhip64: the streams handed to the C-ABI are torch's, so both must share ONE runtime.
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().apexmi_last_error().decode("utf-8", "replace")
        raise ApexMIError(f"{what or 'apexmi'} failed (rc={rc}): {msg}")


def i64x3(vals):
    return (C.c_int64 * 3)(*[int(v) for v in vals])


def tune_set(key: str, value: int) -> None:
    check(load().apexmi_tune_set(key.encode(), int(value)), "tune_set")


def attn_w64_fallbacks() -> int:
    """workgroups of the large-attention kernel that re-ran with the running-maximum loop since the last call (synchronises)"""
    n = C.c_uint64(0)
    check(load().apexmi_attn_w64_fallbacks(C.byref(n)), "attn_w64_fallbacks")
    return int(n.value)


def prof_enable(on: bool) -> None:
    check(load().apexmi_prof_enable(1 if on else 0))


def clk_enable(on: bool) -> None:
    check(load().apexmi_clk_enable(1 if on else 0), "clk_enable")


def clk_read() -> dict:
    """Effective shader clock over the GEMM K-loops since `clk_enable(True)`: summed `s_memtime` cycles / summed 100 MHz ticks."""
    c, r = C.c_uint64(0), C.c_uint64(0)
    check(load().apexmi_clk_read(C.byref(c), C.byref(r)), "clk_read")
    return {"cycles": int(c.value), "ref_ticks": int(r.value), "ghz": (0.1 * c.value / r.value) if r.value else None}


def prof_reset() -> None:
    check(load().apexmi_prof_reset())


def prof_read() -> dict:
    ms = (C.c_double * NCLASS)()
    n = (C.c_int64 * NCLASS)()
    fl = (C.c_double * NCLASS)()
    by = (C.c_double * NCLASS)()
    check(load().apexmi_prof_read(ms, n, fl, by), "prof_read")
    return {
        PROF_CLASSES[i]: {"ms": ms[i], "launches": int(n[i]), "flops": fl[i], "bytes": by[i]}
        for i in range(NCLASS)
    }
