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
