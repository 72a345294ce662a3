#!/usr/bin/env python
"""How does the time of ONE round of 256 x 256 tiles depend on how many CUs hold a tile?  K = 12288 (192 K-tiles: the fixed
per-tile cost is ~5 %), N = 256 x nn, M = 256 x nm with nm x nn = 16 .. 256 tiles, rotating (cold) weights.
If a K-tile's time did not depend on the number of busy CUs the launch time would be flat; if the kernel is bound by a SHARED
resource (L2 -> LDS staging bandwidth, fabric, power) it grows with the tile count.  Prints us per launch, us per K-tile,
aggregate staged TB/s (tiles x 64 KiB per K-tile) and TFLOP/s.  X288=1: the same for the 288 x 192 tiling (gemm.x288 = 2)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
K = int(os.environ.get("K", "12288"))
X288 = os.environ.get("X288", "0") == "1"
X384 = os.environ.get("X384", "0") == "1"          # the 384 x 256 tiling (DIST=0|1: its piece distribution)
BM, BN = (288, 192) if X288 else (384, 256) if X384 else (256, 256)
lib.tune_set("gemm.x288", 2 if X288 else 0)
lib.tune_set("gemm.x384", 2 if X384 else 0)
lib.tune_set("gemm.x384_dist", int(os.environ.get("DIST", "1")))
g = torch.Generator(device=DEV).manual_seed(0)
for nm, nn in [(2, 8), (4, 8), (8, 8), (8, 12), (8, 16), (12, 16), (14, 16), (16, 13), (16, 14), (16, 15), (16, 16), (16, 17), (16, 20), (16, 24), (16, 32)]:
    M, N = BM * nm, BN * nn
    nw = max(2, min(24, int(6e9 // (N * K * 2))))
    ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    reps = 24
    ts = []
    for r in range(3):
        ops.gemm(a, ws[0], None, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ops.gemm(a, ws[i % nw], None, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    us = sorted(ts)[1]
    tiles = nm * nn
    rounds = (tiles + 255) // 256
    per_kt = us / (K / 64) / rounds
    print(json.dumps({"tile": [BM, BN], "tiles": tiles, "M": M, "N": N, "K": K, "us": round(us, 1), "us_per_ktile_round": round(per_kt, 3),
                      "staged_TBps": round(tiles * (BM + BN) * 128 * (K / 64) / us / 1e6, 2),
                      "tflops": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)
    del ws
lib.tune_set("gemm.x288", 0)
lib.tune_set("gemm.x384", 1)
