#!/usr/bin/env python
"""Would warming the Infinity Cache with the NEXT launch's weights (a prefetch stream under the attention kernel, which leaves the
memory system idle) speed the GEMMs up?  Upper bound, measured: every Flux-step GEMM shape timed over 19 launches that each read
their OWN layer's weights (cold: from HBM, as in the step) against 19 launches that re-read ONE layer's weights (warm: activations +
weights + output of a launch fit the 256 MiB Infinity Cache, so every L2 miss of the weight panels is served on-die)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
L = 19
g = torch.Generator(device=DEV).manual_seed(0)


def mk(n, k):
    return (torch.randn(n, k, generator=g, device=DEV) * k ** -0.5).to(torch.bfloat16)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / L * 1e3


S = 4608
tot = {"cold": 0.0, "warm": 0.0}
for name, N, K, per_step in (("double QKV", 9216, 3072, 19), ("double attention-out", 3072, 3072, 19), ("double FF-up", 12288, 3072, 19),
                             ("double FF-down", 3072, 12288, 19), ("single QKV + MLP-up", 21504, 3072, 38), ("single proj_out", 3072, 15360, 38)):
    a = torch.randn(S, K, generator=g, device=DEV).to(torch.bfloat16)
    ws = [mk(N, K) for _ in range(L)]
    out = torch.empty(S, N, device=DEV, dtype=torch.bfloat16)
    res = {"shape": name, "M": S, "N": N, "K": K, "weight_MB": round(N * K * 2 / 1e6, 1), "us": {}}
    for rnd in range(2):
        res["us"].setdefault("cold", []).append(round(timeit(lambda: [ops.gemm(a, w, None, out=out) for w in ws]), 1))
        res["us"].setdefault("warm", []).append(round(timeit(lambda: [ops.gemm(a, ws[0], None, out=out) for _ in ws]), 1))
    c, w = min(res["us"]["cold"]), min(res["us"]["warm"])
    res["warm_gain"] = round(c / w - 1.0, 4)
    tot["cold"] += c * per_step
    tot["warm"] += w * per_step
    print(json.dumps(res), flush=True)
print(json.dumps({"per_step_ms": {k: round(v / 1e3, 2) for k, v in tot.items()}, "upper_bound_gain_ms": round((tot["cold"] - tot["warm"]) / 1e3, 2)}))
