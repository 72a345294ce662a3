#!/usr/bin/env python
"""Stream-K launches (`gemm.streamk` 1 / 0) of the shipped GEMM: agreement with the one-workgroup-per-tile launch (f32 summation order
only: the partial sums of a split tile are added as own range + earlier ranges), determinism over repeated launches (flags lowered by
the owners), ragged and grouped problems, in-place gated residual; then TFLOP/s per Flux shape with cold weights, interleaved."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(torch.bfloat16)


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


def check():
    rep = {}
    # (M, N, K, epilogue): 216 tiles (one part-filled round), 18 x 36 = 648 (two full rounds + 136), ragged M, K = 64 * 8 (smallest)
    for (M, N, K, epi) in [(4608, 3072, 3072, "gate_res"), (4608, 3072, 15360, "gate_res"), (4608, 9216, 3072, "bias"),
                           (4400, 3000, 1024, "gelu"), (4608, 3072, 512, "bias")]:
        a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
        gate = torch.randn(N, generator=g, device=DEV)
        res = rnd(M, N)
        outs = {}
        for sk in (0, 1):
            lib.tune_set("gemm.streamk", sk)
            first = None
            for _ in range(5):
                o = res.clone()
                kw = dict(epilogue=epi)
                if epi == "gate_res":
                    kw.update(gate=gate, residual=o)
                y = ops.gemm(a, w, b, out=o, **kw).clone()
                assert first is None or torch.equal(y, first), f"stream-K={sk}: non-deterministic at {M}x{N}x{K}"
                first = y
            outs[sk] = first.float()
        d = (outs[1] - outs[0]).abs()
        rel = float(d.norm() / outs[0].norm())
        rep[f"{M}x{N}x{K}:{epi}"] = {"rel_l2_vs_tile_launch": rel, "fraction_of_outputs_that_differ": float((d > 0).float().mean())}
        assert rel < 2e-3 and torch.isfinite(outs[1]).all(), (M, N, K, rel)
    # grouped: img + txt streams of a double block's QKV (648 tiles over two problems)
    a1, a2, w1, w2 = rnd(4096, 3072), rnd(512, 3072), rnd(9216, 3072, scale=0.02), rnd(9216, 3072, scale=0.02)
    outs = {}
    for sk in (0, 1):
        lib.tune_set("gemm.streamk", sk)
        o1, o2 = torch.empty(4096, 9216, device=DEV, dtype=torch.bfloat16), torch.empty(512, 9216, device=DEV, dtype=torch.bfloat16)
        ops.gemm_grouped([a1, a2], [w1, w2], [None, None], [o1, o2])
        outs[sk] = torch.cat([o1, o2]).float()
    rep["grouped 4096+512 x 9216 x 3072"] = {"rel_l2_vs_tile_launch": float((outs[1] - outs[0]).norm() / outs[0].norm())}
    assert rep["grouped 4096+512 x 9216 x 3072"]["rel_l2_vs_tile_launch"] < 2e-3
    lib.tune_set("gemm.streamk", 1)
    print(json.dumps({"agreement": rep}), flush=True)


def bench():
    shapes = [("proj_out_single", 4608, 3072, 15360, "gate_res"), ("ff_down_img+txt", 4608, 3072, 12288, "gate_res"),
              ("attn_out_img+txt", 4608, 3072, 3072, "gate_res"), ("qkv_img+txt", 4608, 9216, 3072, "bias"),
              ("ff_up_img+txt", 4608, 12288, 3072, "gelu"), ("qkv_mlp_single", 4608, 21504, 3072, "bias")]
    for name, M, N, K, epi in shapes:
        a = rnd(M, K)
        nw = int(6e8 // (N * K * 2)) + 1
        ws = [rnd(N, K, scale=K ** -0.5) for _ in range(nw)]
        b = rnd(N)
        gate = torch.randn(N, generator=g, device=DEV)
        out = rnd(M, N)
        kw = dict(epilogue=epi)
        if epi == "gate_res":
            kw.update(gate=gate, residual=out)
        res = {0: [], 1: []}
        st = {"i": 0}
        for _ in range(3):
            for sk in res:
                lib.tune_set("gemm.streamk", sk)

                def f():
                    st["i"] = (st["i"] + 1) % nw
                    ops.gemm(a, ws[st["i"]], b, out=out, **kw)
                ms = timeit(f)
                res[sk].append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
        lib.tune_set("gemm.streamk", 1)
        print(json.dumps({"gemm": name, "tiles": ((M + 255) // 256) * ((N + 255) // 256),
                          "tflops": {"tile launch": res[0], "stream-K": res[1]}}), flush=True)
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    check()
    bench()
