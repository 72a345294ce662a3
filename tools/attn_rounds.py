#!/usr/bin/env python
"""Does workgroup-round quantisation cost attention time on a power-limited chip?  Fixed S, head count swept so the
launch is 0.98 / 1.69 / 1.97 / 2.02 / 3.0 ... rounds of 256 workgroups; reports time per workgroup-round-equivalent."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
S = int(os.environ.get("S", 4608))


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


nqb = (S + 255) // 256
skp = (S + 63) // 64 * 64
for H in [int(x) for x in os.environ.get("HEADS", "14,15,24,28,29,32,43,57").split(",")]:
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: ops.attention_prepared(q, k, vt, o, S))
    wgs = nqb * H
    print(json.dumps({"H": H, "S": S, "workgroups": wgs, "rounds": round(wgs / 256, 3), "ms": round(ms, 4),
                      "tflops": round(4.0 * H * S * S * 128 / (ms * 1e-3) / 1e12, 1),
                      "us_per_workgroup_x256": round(ms * 1e3 / wgs * 256, 1)}), flush=True)
