#!/usr/bin/env python
"""The Flux single block's fused QKV + MLP-up launch (M 4608, N 9216 + 12288, K 3072, 24 heads) on the 256 x 256 tiling (gemm.x384 = 0)
and on the 384 x 256 tiling (gemm.x384 = 1), interleaved, rotating weights.  APEX_MI355_LIB=<side build> to compare library variants
(run once per library; the 256 x 256 arm is the common yardstick)."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
H, K, S, mlp = 24, 3072, 4608, 12288
inner = H * 128
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=DEV) * sc).to(torch.bfloat16)   # noqa: E731
x = rn(S, K)
NW = 12
wq, wm = [rn(3 * inner, K, sc=K ** -0.5) for _ in range(NW)], [rn(mlp, K, sc=K ** -0.5) for _ in range(NW)]
bq, bm = rn(3 * inner, sc=0.1), rn(mlp, sc=0.1)
nq, nk = rn(128) * 0.2 + 1, rn(128) * 0.2 + 1
ang = torch.rand(S, 64, generator=g, device=DEV) * 6.283
rope = torch.stack([ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1)]).contiguous().float()
q, k = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
vt = torch.zeros(H, 128, (S + 63) // 64 * 64, device=DEV, dtype=torch.bfloat16)
up = torch.empty(S, mlp, device=DEV, dtype=torch.bfloat16)


def launch(i):
    ops.gemm_grouped_qkv([x, x], [wq[i % NW], wm[i % NW]], [bq, bm], [None, up], ["bias", "gelu"], [1, 0], [nq, None], [nk, None], [0, 0],
                         H, 1e-6, rope, q, k, vt)


res = {0: [], 1: []}
for r in range(5):
    for m in (0, 1):
        lib.tune_set("gemm.x384", m)
        launch(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(38):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / 38 * 1e3)
lib.tune_set("gemm.x384", 1)
med = {m: statistics.median(v) for m, v in res.items()}
print(json.dumps({"lib": os.environ.get("APEX_MI355_LIB", "shipped"), "us_256x256": round(med[0], 1), "us_384x256": round(med[1], 1),
                  "speedup": round(med[0] / med[1], 4)}))
