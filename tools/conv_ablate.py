#!/usr/bin/env python
"""Timing ablation of the prefetch slab kernel (conv.pp = 1): what the loop costs without its DMA (conv.dbg bits: 1 no weight
pieces, 2 no slab pieces inside the loop — results are wrong, only the time is read)."""
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
for name in ("96->96 3x3x3 (full res)", "192->192 3x3x3 (half res)"):
    cin, cout, T, H, W, k, up = CASES[name]
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    res = {}
    for pp in (2, 3):
        lib.tune_set("conv.pp", pp)
        for tag, mask in {"full": 0, "no weight dma": 1, "no slab dma": 2, "no dma": 3}.items():
            lib.tune_set("conv.dbg", mask)
            res[f"pp{pp} {tag}"] = round(min(tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up)) for _ in range(2)), 3)
    lib.tune_set("conv.dbg", 0)
    print(json.dumps({"case": name, "ms": res}), flush=True)
lib.tune_set("conv.pp", 1)
