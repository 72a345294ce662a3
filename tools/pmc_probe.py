#!/usr/bin/env python
"""Few launches of the hot kernels for rocprofv3 --pmc passes (keeps counter collection short)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "gemm"):
    for (M, N, K) in ([(8192, 8192, 8192)] if os.environ.get('GEMM_CFGS') else [(8192, 8192, 8192), (4608, 9216, 3072)]):
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        for cfg, var in [(int(c), 1) for c in os.environ.get('GEMM_CFGS', '2,3').split(',')]:
            lib.tune_set("gemm.config", cfg)
            for _ in range(3):
                ops.gemm(a, w)
    lib.tune_set("gemm.config", 0)
if what in ("all", "attn"):
    H, S = 24, 4608
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.randn(1, H, 128, S, generator=g, device=DEV).to(torch.bfloat16)
    o = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        ops.attention_prepared(q, k, vt, o, S)
torch.cuda.synchronize()
