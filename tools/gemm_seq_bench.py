#!/usr/bin/env python
"""Sustained GEMM throughput on the Flux step's GEMM sequence (cold weights, back-to-back launches for a whole step):
this library's kernels vs torch.matmul (hipBLASLt).  Isolated per-shape numbers are misleading on a power-limited chip;
this is the comparison that matters for the step."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
S = 4608
DOUBLE = [(9216, 3072), (3072, 3072), (12288, 3072), (3072, 12288)]
SINGLE = [(21504, 3072), (3072, 15360)]
g = torch.Generator(device=DEV).manual_seed(0)


def mk(n, k):
    return (torch.randn(n, k, generator=g, device=DEV) * k ** -0.5).to(torch.bfloat16)


layers = []
for _ in range(19):
    layers.append([mk(n, k) for n, k in DOUBLE])
for _ in range(38):
    layers.append([mk(n, k) for n, k in SINGLE])
acts = {k: torch.randn(S, k, generator=g, device=DEV).to(torch.bfloat16) for k in (3072, 12288, 15360)}
outs = {n: torch.empty(S, n, device=DEV, dtype=torch.bfloat16) for n in (9216, 3072, 12288, 21504)}
flops = sum(2.0 * S * w.shape[0] * w.shape[1] for L in layers for w in L)


def run(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ours():
    for L in layers:
        for w in L:
            ops.gemm(acts[w.shape[1]], w, None, out=outs[w.shape[0]])


def blaslt():
    for L in layers:
        for w in L:
            torch.matmul(acts[w.shape[1]], w.t(), out=outs[w.shape[0]])


res = {}
if os.environ.get("SEQ_ONLY"):          # SEQ_ONLY="7,9": just these tilings, no vendor run (ablation builds)
    for cfg in (int(v) for v in os.environ["SEQ_ONLY"].split(",")):
        lib.tune_set("gemm.large", cfg)
        ms = run(ours)
        res[f"cfg{cfg}"] = {"ms_per_step": round(ms, 2), "tflops": round(flops / ms / 1e9, 1)}
    print(json.dumps({"flux_gemm_sequence_tflop": round(flops / 1e12, 2), **res}))
    sys.exit(0)
for name, cfg in (("cfg7", 7), ("cfg9_ring", 9), ("cfg10_ring5", 10), ("cfg7_b", 7), ("cfg9_ring_b", 9), ("cfg10_ring5_b", 10)) if not os.environ.get("SEQ_TUNE") else ():
    lib.tune_set("gemm.large", cfg)
    ms = run(ours)
    res[name] = {"ms_per_step": round(ms, 2), "tflops": round(flops / ms / 1e9, 1)}
lib.tune_set("gemm.large", 7)
for kv in filter(None, os.environ.get("SEQ_TUNE", "").split(";")):     # e.g. SEQ_TUNE="gemm.group_m=4;gemm.group_m=16"
    key, val = kv.split("=")
    lib.tune_set(key, int(val))
    ms = run(ours)
    res[kv] = {"ms_per_step": round(ms, 2), "tflops": round(flops / ms / 1e9, 1)}
lib.tune_set("gemm.group_m", 0)
ms = run(blaslt)
res["torch_matmul"] = {"ms_per_step": round(ms, 2), "tflops": round(flops / ms / 1e9, 1)}
ms = run(ours)
res["cfg7_again"] = {"ms_per_step": round(ms, 2), "tflops": round(flops / ms / 1e9, 1)}
print(json.dumps({"flux_gemm_sequence_tflop": round(flops / 1e12, 2), **res}))
