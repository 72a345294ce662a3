#!/usr/bin/env python
"""EXPERIMENT: operands of the shipped GEMM in TILE-MAJOR packed layout ([rows / 256][K / 64][256][64]: an LDS-DMA piece is 1 KiB
contiguous instead of 8 rows x 128 B at the row stride).  tools/ubench/gemm_roof modes 4 / 5 put the stall-free stream at 1372 TF with
the real row-major addressing and 1474 TF packed (+7.5 %).  Here the real kernel: `gemm.wpacked` = 0 row-major | 1 W packed | 3 both
packed, bit-identity of the results, TFLOP/s per Flux shape with cold weights and the step's whole GEMM sequence, interleaved."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def pack(t):
    """[R, K] -> tile-major [ceil(R / 256), K / 64, 256, 64] (rows zero padded), flattened back to a [R', K] tensor for the wrapper"""
    R, K = t.shape
    Rp = (R + 255) // 256 * 256
    p = torch.zeros(Rp, K, dtype=t.dtype, device=t.device)
    p[:R] = t
    return p.view(Rp // 256, 256, K // 64, 64).permute(0, 2, 1, 3).contiguous().view(Rp, K)[:R]


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
    # bit-identity on a ragged problem
    M, N, K = 1300, 1096, 192
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
    lib.tune_set("gemm.config", 7)
    ref = ops.gemm(a, w, b, epilogue="gelu").clone()
    same = {}
    for mode, (aa, ww) in {1: (a, pack(w)), 2: (pack(a), w), 3: (pack(a), pack(w))}.items():
        lib.tune_set("gemm.wpacked", mode)
        same[mode] = bool(torch.equal(ops.gemm(aa, ww, b, epilogue="gelu"), ref))
    lib.tune_set("gemm.wpacked", 0)
    lib.tune_set("gemm.config", 0)
    print(json.dumps({"bit_identical_to_row_major": same}), flush=True)
    shapes = [("qkv_mlp_single", 4608, 21504, 3072, "bias"), ("proj_out_single", 4608, 3072, 15360, "gate_res"),
              ("ff_down_img", 4096, 3072, 12288, "gate_res"), ("ff_up_img", 4096, 12288, 3072, "gelu"),
              ("attn_out_img", 4096, 3072, 3072, "gate_res"), ("qkv_img", 4096, 9216, 3072, "bias"), ("square_8192", 8192, 8192, 8192, "bias")]
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        ap = pack(a)
        nw = int(6e8 // (N * K * 2)) + 1
        ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
        wps = [pack(w_) for w_ in ws]
        b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
        gate = torch.randn(N, generator=g, device=DEV)
        out = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
        kw = dict(epilogue=epi)
        if epi == "gate_res":
            kw.update(gate=gate, residual=out)
        res = {0: [], 1: [], 3: []}
        st = {"i": 0}
        for _ in range(3):
            for mode in res:
                lib.tune_set("gemm.wpacked", mode)

                def f():
                    st["i"] = (st["i"] + 1) % nw
                    ops.gemm(ap if mode & 2 else a, (wps if mode & 1 else ws)[st["i"]], b, out=out, **kw)
                ms = timeit(f)
                res[mode].append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
        lib.tune_set("gemm.wpacked", 0)
        print(json.dumps({"gemm": name, "tflops": {"row-major": res[0], "W packed": res[1], "A and W packed": res[3]}}), flush=True)
        del ws, wps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
