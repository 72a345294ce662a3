#!/usr/bin/env python
"""Interleaved A/B of the 288 x 192 exact-fill tiling (gemm.x288 = 2) against the shipped 256 x 256 launch (gemm.x288 = 0) on the
shapes where the 256 x 256 tiling leaves a round part-filled: back-to-back launches over ROTATING (cold) weights, as in the step.
  SHAPES="M,N,K,epi;..."   epi = bias | gelu | gate_res          ROUNDS=5  REPS=38
Prints TFLOP/s per arm and shape (median over rounds) and the bit-identity of the outputs."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
DEFAULT = ("4608,3072,15360,gate_res;4096,3072,12288,gate_res;4096,3072,3072,gate_res;8192,3072,3072,gate_res;"
           "8192,3072,12288,gate_res;1536,3072,15360,gate_res;4608,12288,3072,gelu")
SHAPES = [s.split(",") for s in os.environ.get("SHAPES", DEFAULT).split(";") if s]
ROUNDS, REPS = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("REPS", "38"))
ARMS = [int(v) for v in os.environ.get("ARMS", "0,2").split(",")]
for _kv in filter(None, os.environ.get("TUNE", "").split(",")):      # extra keys for the whole run, e.g. TUNE=gemm.x384_dist=0
    lib.tune_set(_kv.split("=")[0], int(_kv.split("=")[1]))
KEY = os.environ.get("KEY", "gemm.x288")        # KEY=gemm.x384: the 384 x 256 tiling
g = torch.Generator(device=DEV).manual_seed(0)
for M, N, K, epi in SHAPES:
    M, N, K = int(M), int(N), int(K)
    nw = min(REPS, max(2, int(6e9 // (N * K * 2))))             # rotating weights: well past the 256 MiB of L2 + MALL
    ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    b = (torch.randn(N, generator=g, device=DEV) * 0.1).to(torch.bfloat16)
    gate = torch.randn(N, generator=g, device=DEV)
    x0 = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(x0)
    uses = lib.load().apexmi_gemm_uses_x288(M, N, K)

    def launch(i):
        if epi == "gate_res":
            ops.gemm(a, ws[i % nw], b, out=out, epilogue="gate_res", gate=gate, residual=x0)
        else:
            ops.gemm(a, ws[i % nw], b, out=out, epilogue=epi)
    res, outs = {m: [] for m in ARMS}, {}
    for r in range(ROUNDS):
        for m in ARMS:
            lib.tune_set(KEY, m)
            launch(0)
            outs.setdefault(m, out.clone())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(REPS):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / REPS * 1e3)
    lib.tune_set(KEY, 1 if KEY == "gemm.x384" else 0)
    fl = 2.0 * M * N * K
    med = {m: statistics.median(v) for m, v in res.items()}
    print(json.dumps({"shape": [M, N, K], "epilogue": epi, "key": KEY, "auto_rule_picks_x288": bool(uses),
                      "us": {str(m): round(v, 1) for m, v in med.items()},
                      "tflops": {str(m): round(fl / v / 1e6, 1) for m, v in med.items()},
                      "x288_speedup": round(med[ARMS[0]] / med[ARMS[-1]], 4),
                      "bit_identical": bool(all(torch.equal(outs[ARMS[0]], o) for o in outs.values()))}), flush=True)
    del ws
    torch.cuda.empty_cache()
