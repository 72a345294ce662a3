#!/usr/bin/env python
"""A/B of the LDS-DMA placement (`attn.dma`) x stage count (`attn.stages`) of the shipped flash-attention kernel on the BASELINE
geometries, interleaved rounds in one process (HIP events); every arm must be bit-identical to the shipped one (2:0)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
SHAPES = {"flux": (24, 4608), "qwen": (24, 8448), "wan": (40, 75600), "long": (8, 32768)}
ARMS = [tuple(int(x) for x in a.split(":")) for a in os.environ.get("ARMS", "2:0,2:4").split(",")]


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
    for rnd in range(int(os.environ.get("ROUNDS", "3"))):
        for arm in ARMS:
            lib.tune_set("attn.stages", arm[0])
            lib.tune_set("attn.xv", arm[1])
            o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
            ms = timeit(lambda: ops.attention_prepared(q, k, vt, o, S), iters)
            best.setdefault(arm, []).append(ms)
            outs[arm] = o
    key = lambda a: f"{a[0]}:{a[1]}"  # noqa: E731
    print(json.dumps({"shape": name, "H": H, "S": S,
                      "ms": {key(a): [round(x, 4) for x in best[a]] for a in ARMS},
                      "tflops": {key(a): round(4.0 * H * S * S * 128 / (min(best[a]) * 1e-3) / 1e12, 1) for a in ARMS},
                      "identical_to_first": {key(a): bool(torch.equal(outs[a], outs[ARMS[0]])) for a in ARMS},
                      "max_abs_diff_vs_first": {key(a): float((outs[a].float() - outs[ARMS[0]].float()).abs().max()) for a in ARMS},
                      "frac_differing": {key(a): float((outs[a] != outs[ARMS[0]]).float().mean()) for a in ARMS}}), flush=True)
lib.tune_set("attn.stages", 2)
lib.tune_set("attn.xv", 0)
