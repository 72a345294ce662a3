#!/usr/bin/env python
"""Time the shipped flash-attention launch at the Flux and a long (Wan-like) shape: TFLOP/s, 20 back-to-back launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
g = torch.Generator(device=DEV).manual_seed(0)
res = {}
for name, H, S in (("flux_24x4608", 24, 4608), ("long_8x32768", 8, 32768)):
    skp = (S + 63) // 64 * 64
    q, k = (torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        ops.attention_prepared(q, k, vt, out, S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    n = 20 if S < 10000 else 5
    e0.record()
    for _ in range(n):
        ops.attention_prepared(q, k, vt, out, S)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[name] = {"ms": round(ms, 3), "tflops": round(4.0 * H * S * S * 128 / ms / 1e9, 1)}
print(json.dumps(res))
