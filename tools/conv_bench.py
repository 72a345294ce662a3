#!/usr/bin/env python
"""Implicit-GEMM convolution on the shapes that hold the Wan / Flux VAE decode's time: v1 (128x128 tile) vs v2 (conv-shaped
tiles), TFLOP/s on algorithmic flops, and bit-equality of the two.  usage: conv_bench.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
# (label, T, H, W, Cin, Cout, ksize, upsample2x)  — one 32x32-latent tile of the 720p x 81f decode, and Flux 1024^2
shapes = [("wan up3 96->96 3x3x3 @81x256x256", 81, 256, 256, 96, 96, (3, 3, 3), False),
          ("diagnostic 128->96 (K-tiles never straddle taps)", 81, 256, 256, 128, 96, (3, 3, 3), False),
          ("diagnostic 64->96", 81, 256, 256, 64, 96, (3, 3, 3), False),
          ("diagnostic 96->192", 81, 128, 128, 96, 192, (3, 3, 3), False),
          ("wan up2 192->192 3x3x3 @81x128x128", 81, 128, 128, 192, 192, (3, 3, 3), False),
          ("wan up2 upsample conv 192->96 3x3 @81x(128->256)", 81, 128, 128, 192, 96, (1, 3, 3), True),
          ("wan up1 384->384 3x3x3 @41x64x64", 41, 64, 64, 384, 384, (3, 3, 3), False),
          ("wan conv_out 96->3 3x3x3 @81x256x256", 81, 256, 256, 96, 3, (3, 3, 3), False),
          ("flux 128->128 3x3 @1024x1024", 1, 1024, 1024, 128, 128, (1, 3, 3), False),
          ("flux 256->256 3x3 @512x512", 1, 512, 512, 256, 256, (1, 3, 3), False),
          ("flux 512->512 3x3 @256x256", 1, 256, 256, 512, 512, (1, 3, 3), False)]
for label, T, H, W, cin, cout, k, up in shapes:
    x = torch.randn(T, H, W, cin, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, *k, generator=g, device=dev) / (cin * k[0] * k[1] * k[2]) ** 0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.zeros(wp.shape[0], dtype=torch.bfloat16, device=dev)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    flops = 2.0 * T * Ho * Wo * cout * cin * k[0] * k[1] * k[2]
    res = {}
    for v2 in (0, 1):
        lib.tune_set("conv.v2", v2)
        out = ops.conv3d_cl(x, wp, b, k, upsample2x=up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = ops.conv3d_cl(x, wp, b, k, upsample2x=up, out=out)
        torch.cuda.synchronize()
        res[v2] = ((time.perf_counter() - t0) / reps, out.clone())
    lib.tune_set("conv.v2", 1)
    same = bool(torch.equal(res[0][1], res[1][1]))
    print(f"{label:52s} v1 {res[0][0] * 1e3:8.3f} ms {flops / res[0][0] / 1e12:7.1f} TF | v2 {res[1][0] * 1e3:8.3f} ms "
          f"{flops / res[1][0] / 1e12:7.1f} TF | x{res[0][0] / res[1][0]:.2f} | bit-equal {same}")
