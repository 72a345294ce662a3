#!/usr/bin/env python
"""The flux 512^2 geometry (S 1024 + 512): which tiling should the part-filled launches run on?  Per shape, interleaved:
gemm.config = 0 (auto: 256 x 256 from 1024 rows up), 8 (128 x 128, eight waves, two workgroups per CU), 1 (128 x 128, four waves).
Rotating (cold) weights, back-to-back launches."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
SHAPES = [(1536, 3072, 15360), (1536, 21504, 3072), (1024, 3072, 12288), (1024, 3072, 3072), (1024, 12288, 3072), (1024, 9216, 3072),
          (512, 3072, 12288), (512, 12288, 3072), (2048, 3072, 12288), (3072, 3072, 12288), (4608, 3072, 15360)]
CFGS = [int(v) for v in os.environ.get("CFGS", "0,8,1").split(",")]
g = torch.Generator(device=DEV).manual_seed(0)
for M, N, K in SHAPES:
    nw = max(2, min(24, int(4e9 // (N * K * 2))))
    ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    res = {c: [] for c in CFGS}
    for r in range(4):
        for c in CFGS:
            lib.tune_set("gemm.config", c)
            ops.gemm(a, ws[0], None, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24):
                ops.gemm(a, ws[i % nw], None, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) / 24 * 1e3)
    lib.tune_set("gemm.config", 0)
    med = {c: statistics.median(v) for c, v in res.items()}
    print(json.dumps({"shape": [M, N, K], "tiles256": ((M + 255) // 256) * ((N + 255) // 256), "us": {str(c): round(v, 1) for c, v in med.items()},
                      "tflops": {str(c): round(2.0 * M * N * K / v / 1e6, 1) for c, v in med.items()}}), flush=True)
    del ws
