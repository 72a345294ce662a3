#!/usr/bin/env python
"""Does the ROW STRIDE of the operands matter to the LDS-DMA staging path?  The ring-schedule ablation (profiles/r04_gemm_ring_ablate.log)
prices the LDS-DMA pieces at ~30 % of the GEMM's time with the step's real addressing, against ~15 % for contiguous 1 KiB pieces in
tools/ubench/gemm_roof.  Here the shipped kernel runs the Flux shapes with operand rows padded by PAD elements (lda = K + PAD), cold
weights, interleaved: A padded / W padded / both / none."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
PADS = [int(v) for v in os.environ.get("PADS", "64,192").split(",")]
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def mk(rows, K, pad, scale=1.0):
    return (torch.randn(rows, K + pad, generator=g, device=DEV) * scale).to(torch.bfloat16)[:, :K]


shapes = [("qkv_mlp_single", 4608, 21504, 3072, "bias"), ("proj_out_single", 4608, 3072, 15360, "gate_res"),
          ("ff_down_img", 4096, 3072, 12288, "gate_res"), ("ff_up_img", 4096, 12288, 3072, "gelu"),
          ("attn_out_img", 4096, 3072, 3072, "gate_res"), ("qkv_img", 4096, 9216, 3072, "bias"), ("square_8192", 8192, 8192, 8192, "bias")]
for name, M, N, K, epi in shapes:
    arms = {"none": (0, 0)}
    for p in PADS:
        arms.update({f"A+{p}": (p, 0), f"W+{p}": (0, p), f"both+{p}": (p, p)})
    ops_ = {}
    for arm, (pa, pw) in arms.items():
        a = mk(M, K, pa)
        nw = max(1, int(6e8 // (N * (K + pw) * 2)))
        ws = [mk(N, K, pw, K ** -0.5) for _ in range(nw + 1)]
        ops_[arm] = (a, ws)
    b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
    gate = torch.randn(N, generator=g, device=DEV)
    out = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
    kw = dict(epilogue=epi)
    if epi == "gate_res":
        kw.update(gate=gate, residual=out)
    res = {k: [] for k in arms}
    for _ in range(3):
        for arm, (a, ws) in ops_.items():
            st = {"i": 0}

            def f():
                st["i"] = (st["i"] + 1) % len(ws)
                ops.gemm(a, ws[st["i"]], b, out=out, **kw)
            ms = timeit(f)
            res[arm].append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
    print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "tflops": res}), flush=True)
    del ops_
    torch.cuda.empty_cache()
