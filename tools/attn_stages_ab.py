#!/usr/bin/env python
"""A/B of the K / V^T LDS ring depth of the shipped attention kernel (`attn.stages` 2 / 3) on the three BASELINE geometries,
interleaved in one process (HIP events), and bit-identity of the two (the same arithmetic on the same tiles)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

SHAPES = {"flux": (24, 4608), "qwen": (24, 8448), "wan": (40, 75600)}


def timeit(fn, iters):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
for name in os.environ.get("SHAPES", "flux,qwen,wan").split(","):
    H, S = SHAPES[name]
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
    outs, best = {}, {}
    iters = 40 if S < 20000 else 3
    for rnd in range(3):
        for ns in (2, 3):
            lib.tune_set("attn.stages", ns)
            ms = timeit(lambda: ops.attention_prepared(q, k, vt, out, S), iters)
            best[ns] = min(best.get(ns, 1e9), ms)
            outs[ns] = out.clone()
    fl = 4.0 * H * S * S * 128
    print(json.dumps({"shape": name, "H": H, "S": S, "ms": {k_: round(v, 4) for k_, v in best.items()},
                      "TFLOPs": {k_: round(fl / v / 1e9, 1) for k_, v in best.items()}, "speedup_3_over_2": round(best[2] / best[3], 4),
                      "bit_identical": bool(torch.equal(outs[2], outs[3]))}), flush=True)
lib.tune_set("attn.stages", 2)
