#!/usr/bin/env python
"""A/B of the slab kernel's schedules (`conv.pp` 0 = serial chunks everywhere, 1 = shipped per-shape choice, 2 = register prefetch everywhere, 3 =
register prefetch with 4 waves of twice the tile where instantiated), interleaved rounds on one box, and bit-identity of all of them (the same sums in the same order)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402
from tools.conv_slab_bench import CASES, tm  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
EXTRA = {"512->512 1x3x3 (flux 256^2, 64-ch slices)": (512, 512, 1, 256, 256, (1, 3, 3), False),
         "128->128 1x3x3 (flux 1024^2)": (128, 128, 1, 1024, 1024, (1, 3, 3), False)}
for name, (cin, cout, T, H, W, k, up) in {**CASES, **EXTRA}.items():
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * T * H * W * (4 if up else 1) * cout * cin * k[0] * 9
    res, outs = {}, {}
    for rnd in range(3):
        for v in (0, 1, 2, 3):
            lib.tune_set("conv.pp", v)
            res.setdefault(v, []).append(round(tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up)), 3))
            outs[v] = ops.conv3d_cl(x, wp, b, k, upsample2x=up)
    print(json.dumps({"case": name, "ms": res, "TFLOPs": {v: round(fl / min(res[v]) / 1e9, 1) for v in res},
                      "speedup": {v: round(min(res[0]) / min(res[v]), 3) for v in (1, 2, 3)},
                      "bit_identical": bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3]))}), flush=True)
lib.tune_set("conv.pp", 1)
