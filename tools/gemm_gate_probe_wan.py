#!/usr/bin/env python
"""The gate / residual epilogue on MANY-round launches (Wan-2.2: 75 600 rows, 384 x 256 tiling): does loading the residual rows in the
epilogue, while the other CUs' K-loops keep the L2 -> CU path saturated, cost what the q/k/v epilogue's table loads cost
(profiles/r06_gemm_qkv_epilogue.log)?  bias vs gate_res, cold weights (8 layers in sequence), plus QwenImage's 8192 + 256-row launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
L = 8
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g, device=DEV) * scale).to(torch.bfloat16)  # noqa: E731


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / L * 1e3


for name, Ms, N, K in (("wan attention-out", [75600], 5120, 5120), ("wan FFN-down", [75600], 5120, 13824),
                       ("qwen attention-out", [8192, 256], 3072, 3072), ("qwen FF-down", [8192, 256], 3072, 12288)):
    acts = [rnd(m, K) for m in Ms]
    ws = [[rnd(N, K, scale=K ** -0.5) for _ in Ms] for _ in range(L)]
    bias = [rnd(N, scale=0.1) for _ in Ms]
    gate = [torch.randn(N, generator=g, device=DEV) for _ in Ms]
    X = [rnd(m, N) for m in Ms]
    out = [torch.empty(m, N, device=DEV, dtype=torch.bfloat16) for m in Ms]

    def plain():
        for l in range(L):
            ops.gemm_grouped(acts, ws[l], bias, out) if len(Ms) > 1 else ops.gemm(acts[0], ws[l][0], bias[0], out=out[0])

    def gated():
        for l in range(L):
            if len(Ms) > 1:
                ops.gemm_grouped(acts, ws[l], bias, X, epilogue="gate_res", gate_list=gate, residual_list=X)
            else:
                ops.gemm(acts[0], ws[l][0], bias[0], out=X[0], epilogue="gate_res", gate=gate[0], residual=X[0])
    res = {"shape": name, "M": Ms, "N": N, "K": K, "us": {}}
    for rnd_ in range(2):
        res["us"].setdefault("bias", []).append(round(timeit(plain), 1))
        res["us"].setdefault("gate_res", []).append(round(timeit(gated), 1))
    b, gr = min(res["us"]["bias"]), min(res["us"]["gate_res"])
    res["extra_us"], res["extra_pct"] = round(gr - b, 1), round(100.0 * (gr / b - 1.0), 2)
    print(json.dumps(res), flush=True)
