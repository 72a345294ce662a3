#!/usr/bin/env python
"""Per-workgroup timeline of the Flux single block's fused launch on the 384 x 256 tiling (needs the -DAPEXMI_GEMM_TRACE=1 side
library: bash tools/gemm_tile_trace.sh build):   APEX_MI355_LIB=tools/ubench/bin/libapex_trace.so python tools/gemm_x384_trace.py
Every workgroup records [kind, entry, K-loop end, epilogue stores acknowledged] (10 ns ticks): medians per kind of tile of the K-loop
and of the epilogue (K-loop end -> all stores acknowledged), for the fused q/k/v + GELU launch and for the plain one."""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
S, K, H, dim, mlp = 4608, 3072, 24, 3072, 12288
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g, device=DEV) * scale).to(torch.bfloat16)  # noqa: E731
a, wq, wm = rnd(S, K), rnd(3 * dim, K, scale=K ** -0.5), rnd(mlp, K, scale=K ** -0.5)
bq, bm, nq, nk = rnd(3 * dim, scale=0.1), rnd(mlp, scale=0.1), rnd(128) * 0.2 + 1, rnd(128) * 0.2 + 1
ang = torch.rand(S, 64, generator=g, device=DEV) * 6.283
rope = torch.stack([ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1)]).contiguous().float()
cat, qkv = torch.empty(S, mlp, device=DEV, dtype=torch.bfloat16), torch.empty(S, 3 * dim, device=DEV, dtype=torch.bfloat16)
Q, Kk = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
VT = torch.zeros(H, 128, S, device=DEV, dtype=torch.bfloat16)
ntiles = 12 * 84


def set_trace(t):
    p = t.data_ptr() if t is not None else 0
    lib.tune_set("gemm.trace_lo", ctypes.c_int32(p & 0xffffffff).value)
    lib.tune_set("gemm.trace_hi", ctypes.c_int32(p >> 32).value)


def run(name, fn):
    tr = torch.zeros(ntiles * 4, dtype=torch.int64, device=DEV)
    fn()
    torch.cuda.synchronize()
    set_trace(tr)
    fn()
    torch.cuda.synchronize()
    set_trace(None)
    r = tr.view(ntiles, 4).cpu()
    out = {"launch_us": round(float(r[:, 3].max() - r[:, 1].min()) / 100.0, 1)}
    for kind, label in ((0, "q"), (1, "k"), (2, "v"), (3, "other")):
        rows = r[r[:, 0] == kind]
        if len(rows):
            out[label] = {"tiles": len(rows), "loop_us": round(statistics.median(((rows[:, 2] - rows[:, 1]).tolist())) / 100.0, 1),
                          "epilogue_us": round(statistics.median(((rows[:, 3] - rows[:, 2]).tolist())) / 100.0, 1),
                          "epilogue_p90_us": round(sorted((rows[:, 3] - rows[:, 2]).tolist())[int(0.9 * len(rows))] / 100.0, 1)}
    print(json.dumps({name: out}), flush=True)


run("fused q/k/v + GELU (the step's launch)",
    lambda: ops.gemm_grouped_qkv([a, a], [wq, wm], [bq, bm], [None, cat], ["bias", "gelu"], [1, 0], [nq, None], [nk, None], [0, 0], H,
                                 1e-6, rope, Q, Kk, VT))
run("bias + GELU, no q/k/v preparation", lambda: ops.gemm_grouped([a, a], [wq, wm], [bq, bm], [qkv, cat], epilogue=["bias", "gelu"]))
