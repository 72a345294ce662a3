"""Build libapex_mi355.so (gfx950) in-tree with hipcc.

The shared library travels to the GPU box with the repo snapshot; nothing is JIT-compiled there.
`python -m apex_studio_amd.build` or `__graft_entry__.build()` call `build()`.
"""
from __future__ import annotations

import hashlib
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


FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("APEXMI_GEMM_STREAMK", "0") not in ("", "0"):
    FLAGS.append("-DAPEXMI_GEMM_STREAMK=1")     # experiment build: the persistent stream-K GEMM launch (gemm.hip; not shipped: slower)
if os.environ.get("APEXMI_DEBUG", "0") not in ("", "0"):
    FLAGS.append("-DAPEXMI_DEBUG")     # experiment knobs of tools/conv_prof.py / conv_ablate.py (apexmi_tune_set "conv.dbg", "conv.prof_*")


# Kernels whose register map is FIXED by hand (one asm statement under a clobber list: attn_w64_body.inc) or that must not touch
# scratch in their loop.  The compiler places the asm's inputs around the clobbered ranges; if an upgrade cannot, it either fails
# the build or spills — the second is silent, so every build parses `-Rpass-analysis=kernel-resource-usage` and refuses a binary
# whose listed kernels use scratch or spill (source file -> substrings of the mangled kernel names).
NO_SPILL = {"attention.hip": ["attn_fwd_d128_w64_kernel", "attn_fwd_d128_w64r_kernel"]}
REMARKS = "-Rpass-analysis=kernel-resource-usage"


def parse_resource_remarks(text: str) -> dict:
    """{mangled kernel name: {"VGPRs": n, "AGPRs": n, "ScratchSize": n, "SGPRs Spill": n, "VGPRs Spill": n, ...}}"""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def check_no_spill(src: str, remarks: str) -> None:
    res = parse_resource_remarks(remarks)
    for want in NO_SPILL.get(src, []):
        hits = {k: v for k, v in res.items() if want in k}
        if not hits:
            raise RuntimeError(f"build: no resource remark for kernel '{want}' of {src} — the spill check cannot run")
        for k, v in hits.items():
            bad = {f: v.get(f, -1) for f in ("ScratchSize", "SGPRs Spill", "VGPRs Spill") if v.get(f, -1) != 0}
            if bad:
                raise RuntimeError(f"build: {k} ({src}) has a hand-fixed register map and must not spill, but the compiler "
                                   f"reports {bad} (VGPRs {v.get('VGPRs')}, AGPRs {v.get('AGPRs')}, SGPRs {v.get('TotalSGPRs')}): "
                                   f"this hipcc allocates differently from the one the map was written against")


def _digest(paths: list[str], extra: str = "") -> str:
    """sha256 over the CONTENT of the inputs (+ compiler flags): mtimes do not survive a repo snapshot / checkout."""
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _stale(target: str, digest: str) -> bool:
    """True unless `target` exists and `target.sha256` records exactly this input digest."""
    stamp = target + ".sha256"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def _stamp(target: str, digest: str) -> None:
    with open(target + ".sha256", "w") as f:
        f.write(digest + "\n")


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"),
               os.path.join(PKG_DIR, "..", "include", "apexmi.h")]
    # generated / multi-include pieces of the translation units (attn_w64_body.inc = tools/gen_attn_w64.py, ...)
    headers += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".inc", ".h")) and f != "common.h")
    objs = []
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(op)
        dg = _digest([sp] + headers, " ".join(FLAGS + ([REMARKS + ":checked"] if src in NO_SPILL else []) + [ARCH]))
        if force or _stale(op, dg):
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *([REMARKS] if src in NO_SPILL else []), "-c", sp, "-o", op]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            # the translation units compile side by side; stderr of the checked ones is parsed (and everything that is not a remark passed on)
            jobs.append((subprocess.Popen(cmd, stderr=subprocess.PIPE if src in NO_SPILL else None, text=True), cmd, op, dg, src))
    for proc, cmd, op, dg, src in jobs:
        err = proc.communicate()[1]          # waits; None unless stderr was piped
        if err:
            rest = [l for l in err.splitlines() if "kernel-resource-usage" not in l and not l.lstrip().startswith(("|", "^"))
                    and not l.strip()[:1].isdigit() and not l.startswith("In file included from")]
            if rest and verbose:
                print("\n".join(rest), file=sys.stderr, flush=True)
        if proc.returncode != 0:
            if err:
                print(err[-4000:], file=sys.stderr)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        if src in NO_SPILL:
            check_no_spill(src, err or "")
        _stamp(op, dg)
    dg = _digest(objs, ARCH)
    if force or _stale(LIB_PATH, dg):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _stamp(LIB_PATH, dg)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
