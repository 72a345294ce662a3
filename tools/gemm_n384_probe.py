#!/usr/bin/env python
"""What would a 256 x 384 tile (256 activation rows x 384 weight rows) buy on the launches of the Flux step that the shipped
384 x 256 tile cannot take (the double blocks' 4096 + 512-row grouped launches, the single block's proj_out)?  Measured WITHOUT
building it: C^T = W A^T is the same tile grid with the operand roles swapped, so the shipped 384 x 256 kernel, forced
(`gemm.x384` = 2) onto the transposed problem (its "activation" = the weight [N, K], its "weight" = the activation [M, K]), stages
exactly the bytes and runs exactly the MFMAs, phases and tile counts a 256 x 384 kernel would on the original one (plain bias-class
epilogue; the transposed store costs the same bytes).  Cold weights: every launch of an arm reads its own layer's weights, 19 layers
in sequence, as in the step.  Arms per shape: shipped path | transposed on the forced 384 x 256 kernel (group heights 3 and 6)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
L = 19
g = torch.Generator(device=DEV).manual_seed(0)


def mk(n, k):
    return (torch.randn(n, k, generator=g, device=DEV) * k ** -0.5).to(torch.bfloat16)


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / L * 1e3      # us per launch


SHAPES = [("double QKV (fused q/k/v epilogue in the step)", [4096, 512], 9216, 3072),
          ("double attention-out", [4096, 512], 3072, 3072),
          ("double FF-up", [4096, 512], 12288, 3072),
          ("double FF-down", [4096, 512], 3072, 12288),
          ("single proj_out", [4608], 3072, 15360),
          ("single QKV + MLP-up (already on 384 x 256)", [4608], 21504, 3072)]
for name, Ms, N, K in SHAPES:
    acts = [torch.randn(m, K, generator=g, device=DEV).to(torch.bfloat16) for m in Ms]
    ws = [[mk(N, K) for _ in Ms] for _ in range(L)]
    outs = [torch.empty(m, N, device=DEV, dtype=torch.bfloat16) for m in Ms]
    outs_t = [torch.empty(N, m, device=DEV, dtype=torch.bfloat16) for m in Ms]

    def shipped():
        for l in range(L):
            if len(Ms) == 1:
                ops.gemm(acts[0], ws[l][0], None, out=outs[0])
            else:
                ops.gemm_grouped(acts, ws[l], None, outs)

    def transposed():
        for l in range(L):
            if len(Ms) == 1:
                ops.gemm(ws[l][0], acts[0], None, out=outs_t[0])
            else:
                ops.gemm_grouped(ws[l], acts, None, outs_t)
    res = {"shape": name, "M": Ms, "N": N, "K": K,
           "tiles_256x256": sum((m + 255) // 256 for m in Ms) * ((N + 255) // 256),
           "tiles_256x384": sum((m + 255) // 256 for m in Ms) * ((N + 383) // 384), "us": {}}
    for rnd in range(2):
        lib.tune_set("gemm.x384", 1)
        res["us"].setdefault("shipped", []).append(round(timeit(shipped), 1))
        for gm in (3, 6):
            lib.tune_set("gemm.x384", 2)
            lib.tune_set("gemm.x384_group_m", gm)
            res["us"].setdefault(f"256x384 emulated, group {gm}", []).append(round(timeit(transposed), 1))
        lib.tune_set("gemm.x384", 1)
        lib.tune_set("gemm.x384_group_m", 3)
    # the transposed arm computes the same numbers
    shipped()
    transposed()
    res["max_abs_diff_vs_transposed"] = max(float((o.float() - t.t().float()).abs().max()) for o, t in zip(outs, outs_t))
    best = {k: min(v) for k, v in res["us"].items()}
    res["gain_of_best_256x384"] = round(best["shipped"] / min(v for k, v in best.items() if k != "shipped") - 1.0, 4)
    print(json.dumps(res), flush=True)
