#!/usr/bin/env python
"""A/B of one integer tune key of the GEMM (`KEY=gemm.phases VALS=4,2`) on the Flux-1024 shapes: bit-identity of the outputs across
the values on ragged / gated problems, then TFLOP/s per shape with cold weights, interleaved rounds.  `gemm.phases` exists only with
profiles/r03_gemm_two_phase_experiment.patch applied; shipped keys work as they are (`KEY=gemm.group_m VALS=6,8`)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
KEY = os.environ.get("KEY", "gemm.phases")
VALS = [int(v) for v in os.environ.get("VALS", "4,2").split(",")]


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


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    same = {}
    for (M, N, K, epi) in [(1024, 1024, 64, "bias"), (1300, 1096, 192, "gelu"), (4608, 3072, 3072, "gate_res"), (1091, 9216, 3072, "bias"),
                           (2048, 1280, 4096, "bias"), (4608, 3072, 15360, "gate_res")]:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
        gate = torch.randn(N, generator=g, device=DEV)
        res = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
        outs = []
        for v in VALS:
            lib.tune_set(KEY, v)
            if KEY != "gemm.config":
                lib.tune_set("gemm.config", 7)
            first = None
            for _ in range(6):
                o = res.clone()
                kw = dict(epilogue=epi)
                if epi == "gate_res":
                    kw.update(gate=gate, residual=o)
                y = ops.gemm(a, w, b, out=o, **kw).clone()
                assert first is None or torch.equal(y, first), "non-deterministic"
                first = y
            outs.append(first)
        same[f"{M}x{N}x{K}:{epi}"] = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
    lib.tune_set("gemm.config", 0)
    if KEY == "gemm.config":          # timing below: the auto path with the tiling chosen by gemm.large
        globals()["KEY"] = "gemm.large"
    lib.tune_set(KEY, VALS[0])
    print(json.dumps({"key": KEY, "values": VALS, "bit_identical_across_values": same}), flush=True)
    shapes = [("qkv_mlp_single", 4608, 21504, 3072, "bias"), ("proj_out_single", 4608, 3072, 15360, "gate_res"),
              ("ff_down_img", 4096, 3072, 12288, "gate_res"), ("ff_up_img", 4096, 12288, 3072, "gelu"),
              ("attn_out_img", 4096, 3072, 3072, "gate_res"), ("qkv_img", 4096, 9216, 3072, "bias"), ("square_8192", 8192, 8192, 8192, "bias")]
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        ws = [w] + [w.clone() for _ in range(int(6e8 // (N * K * 2)))]
        st = {"i": 0}

        def nw():
            st["i"] = (st["i"] + 1) % len(ws)
            return ws[st["i"]]
        b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
        gate = torch.randn(N, generator=g, device=DEV)
        out = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
        kw = dict(epilogue=epi)
        if epi == "gate_res":
            kw.update(gate=gate, residual=out)
        r = {v: [] for v in VALS}
        for _ in range(3):
            for v in VALS:
                lib.tune_set(KEY, v)
                ms = timeit(lambda: ops.gemm(a, nw(), b, out=out, **kw))
                r[v].append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
        lib.tune_set(KEY, VALS[0])
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "tflops": {f"{KEY}={v}": r[v] for v in VALS}}), flush=True)


if __name__ == "__main__":
    main()
