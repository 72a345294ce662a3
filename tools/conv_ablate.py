#!/usr/bin/env python
"""Ablation of the direct-convolution kernel (PP = 0 form): which part of a chunk costs what.  conv.dbg mask bits: 1 no MFMAs,
2 no weight DMA, 4 no slab DMA, 8 no fragment reads, 16 no epilogue, 32 no barriers.  Results are garbage by design."""
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
lib.tune_set("conv.pp", 0)
MASKS = {"full": 0, "no epilogue": 16, "no mfma": 1, "no weight dma": 2, "no slab dma": 4, "no dma": 6, "no reads": 8, "no barrier": 32,
         "mfma only": 2 | 4 | 8 | 16, "mfma + reads": 2 | 4 | 16, "mfma + reads + barrier-free": 2 | 4 | 16 | 32, "dma only": 1 | 8 | 16,
         "dma + reads": 1 | 16, "reads only": 1 | 2 | 4 | 16, "nothing": 1 | 2 | 4 | 8 | 16, "nothing, no barrier": 1 | 2 | 4 | 8 | 16 | 32, "prologue only": 2 | 4 | 64, "prologue with dma": 64, "return at entry": 128}
for name in ("96->96 3x3x3 (full res)", "192->192 3x3x3 (half res)"):
    cin, cout, T, H, W, k, up = CASES[name]
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    res = {}
    for rnd in range(2):
        for tag, mask in MASKS.items():
            lib.tune_set("conv.dbg", mask)
            res.setdefault(tag, []).append(round(tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up)), 3))
    lib.tune_set("conv.dbg", 0)
    print(json.dumps({"case": name, "ms": {t: min(v) for t, v in res.items()}}), flush=True)
