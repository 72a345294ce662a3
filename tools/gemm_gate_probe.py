#!/usr/bin/env python
"""Do the gate / residual epilogues (attention-out, FF-down, proj_out: they LOAD the residual rows) pay the same queueing cost the
q/k/v epilogue's table loads paid (profiles/r06_gemm_qkv_epilogue.log)?  Each Flux shape with the plain bias epilogue against
gate_res, cold weights, 19 layers in sequence; double-block shapes as the grouped [4096 | 512] launch of the step."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
L = 19
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g, device=DEV) * scale).to(torch.bfloat16)  # noqa: E731


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / L * 1e3


for name, Ms, N, K in (("double attention-out", [4096, 512], 3072, 3072), ("double FF-down", [4096, 512], 3072, 12288),
                       ("single proj_out", [4608], 3072, 15360)):
    acts = [rnd(m, K) for m in Ms]
    ws = [[rnd(N, K, scale=K ** -0.5) for _ in Ms] for _ in range(L)]
    bias = [rnd(N, scale=0.1) for _ in Ms]
    gate = [torch.randn(N, generator=g, device=DEV) for _ in Ms]
    X = [rnd(m, N) for m in Ms]                       # the residual stream rows: read and overwritten, as in the step
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
    res = {"shape": name, "us": {}}
    for rnd_ in range(2):
        res["us"].setdefault("bias", []).append(round(timeit(plain), 1))
        res["us"].setdefault("gate_res", []).append(round(timeit(gated), 1))
    res["extra_us"] = round(min(res["us"]["gate_res"]) - min(res["us"]["bias"]), 1)
    print(json.dumps(res), flush=True)
