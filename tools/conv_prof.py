#!/usr/bin/env python
"""Cycle stamps (s_memtime) of one workgroup of the prefetch slab kernel: per wave, cycles per interval spent waiting at the barrier,
issuing the interval's instructions (MFMAs with the next chunk's fragment reads and the DMA pieces in their shadow), in the
vmcnt wait and the lgkmcnt wait.  The stamps are read back after the interval's own lgkmcnt(0), so they add no wait."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402
from tools.conv_slab_bench import CASES, tm  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
buf = torch.zeros(64, dtype=torch.int64, device=DEV)
p = buf.data_ptr()
lo, hi = p & 0xffffffff, p >> 32
lib.tune_set("conv.prof_lo", ctypes.c_int32(lo).value)
lib.tune_set("conv.prof_hi", ctypes.c_int32(hi).value)
lib.tune_set("conv.pp", 2)
for name in ("96->96 3x3x3 (full res)", "192->192 3x3x3 (half res)"):
    cin, cout, T, H, W, k, up = CASES[name]
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    ms = tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up))
    torch.cuda.synchronize()
    r = buf.view(8, 8).cpu().tolist()
    n = max(r[0][5], 1)
    rows = {f"wave{w_}": {"barrier": round(r[w_][0] / n), "issue(reads+mfma+dma)": round(r[w_][1] / n), "vmcnt": round(r[w_][2] / n),
                          "lgkm": round(r[w_][3] / n), "sum": round(sum(r[w_][:4]) / n)} for w_ in range(8)}
    print(json.dumps({"case": name, "ms_with_stamps": round(ms, 3), "chunks": n, "cycles_per_chunk": rows}), flush=True)
lib.tune_set("conv.prof_lo", 0)
lib.tune_set("conv.prof_hi", 0)
lib.tune_set("conv.pp", 1)
