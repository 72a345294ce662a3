#!/usr/bin/env python
"""A/B of the temporal-tap order of the slab kernels (`conv.torder` 0 = oldest frame first, 1 = by input frame mod 3 with the
frame-fastest tile order), interleaved rounds, and how far the two results are apart (f32 summation order over the temporal taps)."""
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
for name, (cin, cout, T, H, W, k, up) in CASES.items():
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    res, outs = {}, {}
    for rnd in range(3):
        for v in (0, 1):
            lib.tune_set("conv.torder", v)
            res.setdefault(v, []).append(round(tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up)), 3))
            outs[v] = ops.conv3d_cl(x, wp, b, k, upsample2x=up).float()
    d = outs[1] - outs[0]
    print(json.dumps({"case": name, "ms": res, "speedup": round(min(res[0]) / min(res[1]), 3),
                      "rel_l2": float(d.norm() / outs[0].norm()), "elements_differing": float((d != 0).float().mean())}), flush=True)
lib.tune_set("conv.torder", 1)
