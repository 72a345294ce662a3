#!/usr/bin/env python
"""One convolution case of tools/conv_slab_bench.CASES run a few times (for rocprofv3 --pmc passes): CONV_CASE = substring of the case
name, CONV_PP = conv.pp value, CONV_N = launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402
from tools.conv_slab_bench import CASES  # noqa: E402

pick = os.environ.get("CONV_CASE", "96->96")
name = [n for n in CASES if pick in n][0]
cin, cout, T, H, W, k, up = CASES[name]
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(T, H, W, cin, generator=g, device="cuda").to(torch.bfloat16)
w = (torch.randn(cout, cin, *k, generator=g, device="cuda") * (cin * k[0] * 9) ** -0.5).to(torch.bfloat16)
wp = ops.pack_conv_weight(w)
b = torch.zeros(wp.shape[0], device="cuda", dtype=torch.bfloat16)
lib.tune_set("conv.pp", int(os.environ.get("CONV_PP", "1")))
for _ in range(int(os.environ.get("CONV_N", "6"))):
    ops.conv3d_cl(x, wp, b, k, upsample2x=up)
torch.cuda.synchronize()
print(name, "done")
