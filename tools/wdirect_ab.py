#!/usr/bin/env python
"""The 256x256 GEMM with the weight fragments loaded straight into registers (`gemm.wdirect=1`, SCHED 7 of gemm.hip) against the
shipped LDS-staged schedule: bit-identity on ragged / grouped / fused-QKV problems, then TFLOP/s per Flux shape (cold weights,
interleaved rounds).  NEEDS profiles/r03_gemm_weights_direct_experiment.patch applied (the schedule is not in the shipped
library: `git apply profiles/r03_gemm_weights_direct_experiment.patch && python -c "import __graft_entry__ as g; g.build()"`)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"


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


def pack_w(w):
    """[N, K] -> [N/16][K/32] blocks of 1 KiB in MFMA operand order (lane l15 + 16 g4 holds row l15, k 8 g4..+7), as [N, K]."""
    N, K = w.shape
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    same = {}
    for (M, N, K, epi) in [(1024, 1024, 64, "bias"), (1300, 1096, 192, "gelu"), (4608, 3072, 3072, "gate_res"), (1091, 9216, 3072, "bias"),
                           (2048, 1280, 4096, "bias")]:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
        gate = torch.randn(N, generator=g, device=DEV)
        res = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
        outs = []
        for wd in (0, 1):
            lib.tune_set("gemm.wdirect", wd)
            lib.tune_set("gemm.config", 7)
            kw = dict(epilogue=epi)
            o = res.clone()
            if epi == "gate_res":
                kw.update(gate=gate, residual=o)
            outs.append(ops.gemm(a, w, b, out=o, **kw).clone())
            for _ in range(5):
                o2 = res.clone()
                if epi == "gate_res":
                    kw.update(residual=o2)
                assert torch.equal(ops.gemm(a, w, b, out=o2, **kw), outs[-1]), "non-deterministic"
        ref = a.float() @ w.float().T + b.float()
        same[f"{M}x{N}x{K}:{epi}"] = bool(torch.equal(outs[0], outs[1]))
        if epi == "bias":
            err = float((outs[1].float() - ref).norm() / ref.norm())
            assert err < 3e-3, err
    lib.tune_set("gemm.config", 0)
    print(json.dumps({"bit_identical_to_lds_staged": same}), flush=True)
    a = torch.randn(1091, 3072, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(9216, 3072, generator=g, device=DEV) * 3072 ** -0.5).to(torch.bfloat16)
    b = torch.randn(9216, generator=g, device=DEV).to(torch.bfloat16)
    lib.tune_set("gemm.wdirect", 0)
    o0 = ops.gemm(a, w, b).clone()
    lib.tune_set("gemm.wdirect", 2)
    o2 = ops.gemm(a, pack_w(w), b).clone()
    lib.tune_set("gemm.wdirect", 3)
    o3 = ops.gemm(a, w, b).clone()
    lib.tune_set("gemm.wdirect", 4)
    o4 = ops.gemm(a, pack_w(w), b).clone()
    rep = all(torch.equal(ops.gemm(a, pack_w(w), b), o4) for _ in range(8))
    lib.tune_set("gemm.wdirect", 0)
    print(json.dumps({"packed_bit_identical": bool(torch.equal(o0, o2)), "landing_bit_identical": bool(torch.equal(o0, o3)),
                      "landing_packed_bit_identical": bool(torch.equal(o0, o4)), "repeatable": bool(rep)}), flush=True)
    shapes = [("qkv_mlp_single", 4608, 21504, 3072, "bias"), ("proj_out_single", 4608, 3072, 15360, "gate_res"),
              ("ff_down_img", 4096, 3072, 12288, "gate_res"), ("ff_up_img", 4096, 12288, 3072, "gelu"),
              ("attn_out_img", 4096, 3072, 3072, "gate_res"), ("square_8192", 8192, 8192, 8192, "bias")]
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        ws = [w] + [w.clone() for _ in range(int(6e8 // (N * K * 2)))]
        wsp = [pack_w(x) for x in ws]
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
        r = {0: [], 1: [], 2: [], 3: [], 4: []}
        for _ in range(3):
            for wd in (0, 1, 2, 3, 4):
                lib.tune_set("gemm.wdirect", wd)
                src = wsp if wd in (2, 4) else ws

                def nw():
                    st["i"] = (st["i"] + 1) % len(src)
                    return src[st["i"]]
                ms = timeit(lambda: ops.gemm(a, nw(), b, out=out, **kw))
                r[wd].append(round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1))
        lib.tune_set("gemm.wdirect", 0)
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "tflops": {"lds_staged": r[0], "weights_direct": r[1], "weights_direct_packed": r[2], "landing_regs": r[3], "landing_regs_packed": r[4]}}), flush=True)


if __name__ == "__main__":
    main()
