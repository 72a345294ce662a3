#!/usr/bin/env python
"""ln_modulate at the Flux shape (4608 x 3072 bf16, two modulation sets) against a plain copy of the same bytes: rows per wave
(`ln.wave` 1 / 2), the row-per-workgroup kernel (0), torch's copy kernel as the floor a 56.6 MB pass has on this chip.  Buffers
rotate over 4 (113 MB: MALL-warm, as behind the producing GEMM in the step)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
S, C, NB = 4608, 3072, 4
g = torch.Generator(device=DEV).manual_seed(0)
xs = [torch.randn(S, C, generator=g, device=DEV).to(torch.bfloat16) for _ in range(NB)]
outs = [torch.empty_like(x) for x in xs]
sc, sh, sc2, sh2 = (torch.randn(C, generator=g, device=DEV) * 0.1 for _ in range(4))


def timeit(fn, iters=200, warm=20):
    for i in range(warm):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


res = {}
ref = None
for rnd in range(3):
    for mode in (1, 2, 0):
        lib.tune_set("ln.wave", mode)
        us = timeit(lambda i: ops.ln_modulate(xs[i % NB], sc, sh, out=outs[i % NB], split=512, scale2=sc2, shift2=sh2))
        res.setdefault(f"ln.wave={mode}", []).append(us)
        o = outs[0].clone()
        if mode == 1 and ref is None:
            ref = o
        elif mode == 2:
            assert torch.equal(o, ref), "ln.wave=2 must be bit-identical to ln.wave=1"
    res.setdefault("torch copy", []).append(timeit(lambda i: outs[i % NB].copy_(xs[i % NB])))
lib.tune_set("ln.wave", 1)
by = 2.0 * S * C * 2
print(json.dumps({"shape": [S, C], "bytes": by, "us": {k: [round(v, 2) for v in vs] for k, vs in res.items()},
                  "TBps_best": {k: round(by / min(vs) / 1e6, 2) for k, vs in res.items()}}))
