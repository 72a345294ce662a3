"""The C-ABI library builds, loads and exports every symbol include/apexmi.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "apexmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(apexmi_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("apexmi_attn_fwd", "apexmi_gemm_bf16", "apexmi_ln_modulate", "apexmi_qkv_prepare",
                 "apexmi_gemv", "apexmi_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        from apex_studio_amd import build
        build.build(verbose=False)
    handle = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(handle, n)]
    assert not missing, f"declared in apexmi.h but not exported: {missing}"
    # and the Python binding table covers the header one-to-one
    assert sorted(lib.SIGNATURES) == _declared()
    assert lib.load().apexmi_version() >= 100


def test_product_ops_refuse_cpu_tensors():
    import torch
    from apex_studio_amd import ops, lib
    a = torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(lib.ApexMIError):
        ops.gemm(a, a)
    with pytest.raises(lib.ApexMIError):
        ops.attention(torch.zeros(1, 2, 8, 64), torch.zeros(1, 2, 8, 64), torch.zeros(1, 2, 8, 64))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from apex_studio_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.ApexMIError):
        lib.load()


def test_argument_validation_is_host_side_and_reports_a_reason():
    """Every entry point validates its arguments before touching the device and leaves a message in
    apexmi_last_error(): exercised here without a GPU, with dummy (never dereferenced) pointers."""
    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lib
    L = lib.load()
    P = 0x100000                      # 16-byte aligned dummy address
    i3 = lib.i64x3((0, 0, 0))

    def bad(rc, needle):
        msg = L.apexmi_last_error().decode()
        assert rc != 0 and needle in msg, (rc, msg)

    bad(L.apexmi_gemm_bf16(P, 64, P, 64, None, P, 64, 8, 8, 100, 0, None, None, 0, None), "K=100")
    bad(L.apexmi_gemm_bf16(P, 64, P, 64, None, P, 64, 8, 12, 64, 0, None, None, 0, None), "N=12")
    bad(L.apexmi_gemm_bf16(P, 64, P, 64, None, P, 64, 8, 8, 64, 9, None, None, 0, None), "epilogue")
    bad(L.apexmi_gemm_bf16(P, 64, P, 64, None, P, 64, 8, 8, 64, lib.EPI_BIAS_GATE_RES, None, None, 0, None), "gate")
    # (round 6: the K-loops' LDS-DMA pieces carry 32-bit lane offsets inside a tile)
    bad(L.apexmi_gemm_bf16(P, (1 << 22) + 64, P, 64, None, P, 64, 2048, 2048, 256, 0, None, None, 0, None), "leading dimensions above 2^22")
    bad(L.apexmi_gemm_bf16(P, 64, P, (1 << 22) + 64, None, P, 64, 8, 8, 64, 0, None, None, 0, None), "leading dimensions above 2^22")
    bad(L.apexmi_gemm_bf16_batched(P, 64, 64, P, 64, 64, P, 64, 64, 0, 8, 8, 64, 0, None), "batch=0")
    bad(L.apexmi_gemm_bf16_batched(P, 64, 64, P, 64, 64, P, 64, 64, 2, 8, 8, 64, lib.EPI_BIAS_GELU, None), "epilogue")
    bad(L.apexmi_attn_fwd_bias(P, 64, P, 64, P, 64, P, 64, 4, 3, 8, 8, 128, 1.0, None, None, None, 0, P, 1 << 30, None),
        "key/value heads")
    bad(L.apexmi_attn_fwd_bias(P, 64, P, 64, P, 64, P, 64, 2, 2, 8, 8, 80, 1.0, None, None, None, 0, P, 1 << 30, None), "head dim 80")
    bad(L.apexmi_attn_fwd_bias(P, 64, P, 64, P, 64, P, 64, 2, 2, 8, 16, 128, 1.0, None, None, None, 1, P, 1 << 30, None), "Sq == Sk")
    bad(L.apexmi_attn_fwd_bias(P, 64, P, 64, P, 64, P, 64, 2, 2, 8, 8, 128, 1.0, None, None, None, 0, P, 16, None), "workspace too small")
    bad(L.apexmi_attn_fwd_framecausal(P, P, P, P, 1, 1, 100, 128, 30, i3, i3, i3, i3, 1.0, P, 1 << 30, None), "whole number of frames")
    bad(L.apexmi_attn_fwd_framecausal(P, P, P, P, 1, 1, 100, 192, 50, i3, i3, i3, i3, 1.0, P, 1 << 30, None), "multiple of 128")
    bad(L.apexmi_conv3d_cl_replicate(P, P, None, None, P, P, 2, 4, 4, 12, 8, 384, 3, 3, 3, None), "Cin=12")
    bad(L.apexmi_rmsnorm_cl(P, P, P, 10, 2048, 0, None), "C=2048")
    bad(L.apexmi_rope_half(P, 256, 4, 2, 64, 80, P, P, None), "head_stride")
    bad(L.apexmi_add_bf16(P, P, P, 12, None), "n=12")
    bad(L.apexmi_mul_bf16(P, P, P, 0, None), "n=0")
    bad(L.apexmi_gather_rows_bf16(P, 100, 10, P, None, 0, 1, P, 128, 4, 100, None), "C=100")
    bad(L.apexmi_frames_to_u8(P, 1, 1, 1, 1, 5, 1, 4, 4, P, None), "C=5")
    bad(L.apexmi_relpos_bias(P, 32, 0, P, 4, 4, P, None), "bad arguments")
    bad(L.apexmi_qkv_prepare(P, P, P, 384, 8, 3, 64, 0, None, None, None, None, 1e-6, None, lib.ROPE_NONE, P, P, P, 8, 64, 0, None), "D=64")
    bad(L.apexmi_tune_set(b"no.such.key", 1), "")
    assert L.apexmi_attn_bias_workspace_bytes(2, 8, 8, 128) > 0 and L.apexmi_attn_framecausal_workspace_bytes(64, 128) > 0
