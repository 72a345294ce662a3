#!/usr/bin/env python
"""Round quantisation of the 256x256 GEMM tiling: N = 12288, K = 3072 (the MM-DiT feed-forward up-projection), M swept so
the launch is 3.0 ... 6.2 rounds of 256 tiles; cold weights."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
N, K = int(os.environ.get("N", 12288)), int(os.environ.get("K", 3072))


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


ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(9)]
b = torch.randn(N, device=DEV).to(torch.bfloat16)
for M in [int(x) for x in os.environ.get("MS", "4096,4352,4608,5120,8192,8448,8704").split(",")]:
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    i = [0]

    def f():
        i[0] = (i[0] + 1) % len(ws)
        ops.gemm(a, ws[i[0]], b, out=out, epilogue="gelu")
    ms = timeit(f)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(json.dumps({"M": M, "tiles": tiles, "rounds": round(tiles / 256, 3), "us": round(ms * 1e3, 1),
                      "tflops": round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1),
                      "us_per_round_equiv": round(ms * 1e3 / (tiles / 256), 1)}), flush=True)
