#!/usr/bin/env python
"""Fixed cost of a GEMM launch on the step's tilings: time(K) = a + b K for the Flux shapes' (M, N), back-to-back launches over
rotating weights (no L2 / MALL reuse between launches).  a = what a tile pays besides its K-loop (workgroup launch, prologue until
the first fragments are in LDS, epilogue, the launch's ramp and tail); b = the K-loop.  Short-K GEMMs (K = 3072: 61 % of the step's GEMM
flops) run at 1000-1050 TFLOP/s against 1300 for K = 15360; this prices the difference."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
SHAPES = {"ff_up (4608 x 12288)": (4608, 12288), "attn_out (4608 x 3072)": (4608, 3072), "qkv_mlp_single (4608 x 21504)": (4608, 21504)}
KS = [256, 512, 1024, 2048, 3072, 6144, 12288]
NW = 6
for name, (M, N) in SHAPES.items():
    rows = []
    for K in KS:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(NW)]
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        for w in ws:
            ops.gemm(a, w, None, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            for w in ws:
                ops.gemm(a, w, None, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (reps * NW)
        rows.append((K, us, 2.0 * M * N * K / us / 1e6))
        del ws, a
    # least squares over K >= 1024
    xs = [r[0] for r in rows if r[0] >= 1024]
    ys = [r[1] for r in rows if r[0] >= 1024]
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    a0 = my - b * mx
    print(json.dumps({"shape": name, "us_by_K": {r[0]: round(r[1], 1) for r in rows}, "tflops_by_K": {r[0]: round(r[2]) for r in rows},
                      "fixed_us": round(a0, 1), "us_per_64_deep_k_tile": round(b * 64, 3),
                      "asymptotic_tflops": round(2.0 * M * N / b / 1e6), "fixed_share_at_K3072": round(a0 / (a0 + b * 3072), 3)}), flush=True)
