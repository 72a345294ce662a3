#!/usr/bin/env python
"""Per-shape A/B of the GEMM tilings and the attention kernel on the Flux-1024 shapes (HIP events,
random bf16 data, interleaved rounds in one process).  Writes JSON lines to stdout."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
CFGS = tuple(int(c) for c in os.environ.get("GEMM_CFGS", "3,6,7").split(","))
SHAPES_ALL = [  # (name, M, N, K, epilogue)
    ("qkv_joint", 4608, 9216, 3072, "bias"), ("mlp_up_single", 4608, 12288, 3072, "gelu"),
    ("proj_out_single", 4608, 3072, 15360, "gate_res"), ("attn_out_img", 4096, 3072, 3072, "gate_res"),
    ("ff_down_img", 4096, 3072, 12288, "gate_res"), ("ff_up_img", 4096, 12288, 3072, "gelu"),
    ("qkv_txt", 512, 9216, 3072, "bias"), ("qwen_ff_up_txt", 256, 12288, 3072, "gelu"), ("qwen_ff_down_txt", 256, 3072, 12288, "gate_res"),
    ("t_4608_3072_12288", 4608, 3072, 12288, "gate_res"), ("t_4096_3072_15360", 4096, 3072, 15360, "gate_res"),
    ("t_4096_3072_8192", 4096, 3072, 8192, "gate_res"), ("t_4096_3072_12352", 4096, 3072, 12352, "gate_res"),
    ("t_4096_4096_12288", 4096, 4096, 12288, "gate_res"), ("square_4096", 4096, 4096, 4096, "bias"),
    ("square_8192", 8192, 8192, 8192, "bias"),
]


SHAPES = [x for x in SHAPES_ALL if not os.environ.get('GEMM_SHAPES') or x[0] in os.environ['GEMM_SHAPES'].split(',')]


def timeit(fn, iters=int(os.environ.get('GEMM_ITERS', '20')), warm=3):
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
    for name, M, N, K, epi in SHAPES:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
        # GEMM_COLD=1: cycle through enough weight copies (> 600 MB) that W comes from HBM, as in the model
        ws = [w] + [w.clone() for _ in range(int(6e8 // (N * K * 2)))] if os.environ.get("GEMM_COLD") else [w]
        state = {"i": 0}

        def next_w():
            state["i"] = (state["i"] + 1) % len(ws)
            return ws[state["i"]]
        b = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
        gate = torch.randn(N, generator=g, device=DEV)
        out = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
        res = {}
        for rnd in range(2):
            for cfg in CFGS:
                lib.tune_set("gemm.config", cfg)
                kw = dict(epilogue=epi)
                if epi == "gate_res":
                    kw.update(gate=gate, residual=out)
                ms = timeit(lambda: ops.gemm(a, next_w(), b, out=out, **kw))
                res.setdefault(cfg, []).append(2.0 * M * N * K / (ms * 1e-3) / 1e12)
        lib.tune_set("gemm.config", 0)
        ref_ms = timeit(lambda: torch.matmul(a, w.t()))
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "epi": epi,
                          "tflops": {f"cfg{c}": [round(x, 1) for x in v] for c, v in res.items()},
                          "torch_matmul_tflops": round(2.0 * M * N * K / (ref_ms * 1e-3) / 1e12, 1)}), flush=True)
    for (H, S) in ([] if os.environ.get('GEMM_SHAPES', '') not in ('', 'attn') else [(24, 4608), (24, 1536), (40, 8192)]):
        q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
        k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
        vt = torch.randn(1, H, 128, S, generator=g, device=DEV).to(torch.bfloat16)
        o = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
        res = {}
        for w in (16, 32):
            lib.tune_set("attn.mfma", w)
            res[w] = round(4.0 * H * S * S * 128 / (timeit(lambda: ops.attention_prepared(q, k, vt, o, S)) * 1e-3) / 1e12, 1)
        lib.tune_set("attn.mfma", 32)
        for c4 in (0, 2):
            lib.tune_set("attn.c4", c4)
            res[f"c4_{c4}"] = round(4.0 * H * S * S * 128 / (timeit(lambda: ops.attention_prepared(q, k, vt, o, S)) * 1e-3) / 1e12, 1)
        lib.tune_set("attn.c4", 3)
        ms = timeit(lambda: ops.attention_prepared(q, k, vt, o, S))
        v = vt.transpose(2, 3).contiguous()
        ref_ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), iters=5)
        print(json.dumps({"attention": [H, S], "tflops": round(4.0 * H * S * S * 128 / (ms * 1e-3) / 1e12, 1), "by_mfma": res,
                          "torch_sdpa_tflops": round(4.0 * H * S * S * 128 / (ref_ms * 1e-3) / 1e12, 1)}), flush=True)


if __name__ == "__main__":
    main()
