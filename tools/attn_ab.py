#!/usr/bin/env python
"""A/B of the flash-attention variants (`attn.c4` = 1 + bits: 1 s_setprio, 2 packed-f32 softmax (shipped = 3), 4 static priority of waves 4..7) on the
three BASELINE geometries, interleaved in one process (HIP events), plus the largest element difference between variants."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
SHAPES = {"flux": (24, 4608), "qwen": (24, 8448), "wan": (40, 75600)}
VARIANTS = [int(x) for x in os.environ.get("VARIANTS", "1,3,5,7,2").split(",")]


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


for name in os.environ.get("SHAPES", "flux,qwen,wan").split(","):
    H, S = SHAPES[name]
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    outs, best = {}, {}
    iters = 40 if S < 20000 else 3
    for rnd in range(3):
        for v in VARIANTS:
            lib.tune_set("attn.c4", v)
            o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
            ms = timeit(lambda: ops.attention_prepared(q, k, vt, o, S), iters)
            best.setdefault(v, []).append(ms)
            outs[v] = o
    ref = outs[VARIANTS[0]].float()
    print(json.dumps({"shape": name, "H": H, "S": S,
                      "ms": {v: [round(x, 4) for x in best[v]] for v in VARIANTS},
                      "tflops": {v: round(4.0 * H * S * S * 128 / (min(best[v]) * 1e-3) / 1e12, 1) for v in VARIANTS},
                      "max_abs_diff_vs_first": {v: float((outs[v].float() - ref).abs().max()) for v in VARIANTS},
                      "frac_differing": {v: float((outs[v] != outs[VARIANTS[0]]).float().mean()) for v in VARIANTS}}), flush=True)
lib.tune_set("attn.c4", 3)
