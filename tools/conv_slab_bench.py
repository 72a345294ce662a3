#!/usr/bin/env python
"""Time the Wan VAE decoder's big convolutions on the direct (slab) kernels against the implicit-GEMM tiles (`conv.slab` 2 / 0)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)


def tm(fn, it=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


CASES = {   # cin, cout, T, H, W, k, upsample: one 32x32-latent tile of the 720p decode
    "96->96 3x3x3 (full res)": (96, 96, 81, 256, 256, (3, 3, 3), False),
    "96->3 conv_out": (96, 3, 81, 256, 256, (3, 3, 3), False),
    "192->192 3x3x3 (half res)": (192, 192, 81, 128, 128, (3, 3, 3), False),
    "192->96 up2 1x3x3": (192, 96, 81, 128, 128, (1, 3, 3), True),
    "384->384 3x3x3 (quarter res)": (384, 384, 41, 64, 64, (3, 3, 3), False),
    "384->192 up2 1x3x3": (384, 192, 41, 64, 64, (1, 3, 3), True),
}
variants = [int(v) for v in os.environ.get("VARIANTS", "0,2").split(",")]
for name, (cin, cout, T, H, W, k, up) in CASES.items():
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=DEV) * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * T * H * W * (4 if up else 1) * cout * cin * k[0] * 9
    res = {}
    for rnd in range(2):
        for v in variants:
            lib.tune_set("conv.slab", v)
            res.setdefault(v, []).append(round(tm(lambda: ops.conv3d_cl(x, wp, b, k, upsample2x=up)), 3))
    print(json.dumps({"case": name, "ms": res, "TFLOPs": {v: round(fl / min(res[v]) / 1e9, 1) for v in res}}), flush=True)
lib.tune_set("conv.slab", 2)
