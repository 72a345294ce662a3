"""Build libapex_mi355.so (gfx950) in-tree with hipcc.

The shared library travels to the GPU box with the repo snapshot; nothing is JIT-compiled there.
`python -m apex_studio_amd.build` or `__graft_entry__.build()` call `build()`.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libapex_mi355.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
SOURCES = ["runtime.hip", "gemm.hip", "attention.hip", "elementwise.hip", "conv.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"),
               os.path.join(PKG_DIR, "..", "include", "apexmi.h")]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", op,
                   "-Wno-unused-result"]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
