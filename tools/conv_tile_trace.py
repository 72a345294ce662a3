#!/usr/bin/env python
"""Per-workgroup timeline of the direct-convolution kernels on the Wan decoder's full-resolution layer shapes (one 256 x 256-pixel
tile of the tiled decode: [T 21.., H 256, W 256]); side library built with -DAPEXMI_CONV_TRACE=1 (bash tools/gemm_tile_trace.sh build):
    APEX_MI355_LIB=tools/ubench/bin/libapex_trace.so python tools/conv_tile_trace.py"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def run(name, T, H, W, cin, cout, norm, res):
    x = torch.randn(T, H, W, cin, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g, device=DEV) * (27 * cin) ** -0.5).to(torch.bfloat16)
    wp = ops.pack_conv_weight(w)
    b = torch.randn(wp.shape[0], generator=g, device=DEV).to(torch.bfloat16) * 0.02
    gam = (1 + 0.1 * torch.randn(wp.shape[0], generator=g, device=DEV)).to(torch.bfloat16)
    r = torch.randn(T, H, W, wp.shape[0], generator=g, device=DEV).to(torch.bfloat16) if res else None

    def call():
        if norm:
            return ops.conv3d_cl_norm(x, wp, b, (3, 3, 3), gam, silu=True, residual=r, want_raw=True)
        return ops.conv3d_cl(x, wp, b, (3, 3, 3), residual=r)
    call()
    torch.cuda.synchronize()
    n = 65536
    tr = torch.zeros(n * 8, dtype=torch.int64, device=DEV)
    os.environ["APEXMI_CONV_TRACE_PTR"] = hex(tr.data_ptr())
    call()
    torch.cuda.synchronize()
    os.environ.pop("APEXMI_CONV_TRACE_PTR")
    rr = tr.view(n, 8).cpu()
    rr = rr[rr[:, 2] != 0]
    key = ((rr[:, 1] & 0xf) << 16) | (rr[:, 0] & 0xff00)
    per_cu = {}
    for i in range(rr.shape[0]):
        per_cu.setdefault(int(key[i]), []).append([int(v) for v in rr[i, 2:7]])
    gaps, pro, loop, epi, drain = [], [], [], [], []
    t0 = min(v[0] for rows in per_cu.values() for v in rows)
    t1 = max(v[4] for rows in per_cu.values() for v in rows)
    for rows in per_cu.values():
        rows.sort()
        for j, (t_in, l0, l1, st, ack) in enumerate(rows):
            pro.append(l0 - t_in)
            loop.append(l1 - l0)
            epi.append(st - l1)
            drain.append(ack - st)
            if j:
                gaps.append(t_in - rows[j - 1][4])
    us = lambda v: round(statistics.median(v) / 100.0, 2) if v else None  # noqa: E731
    print(json.dumps({"conv": name, "workgroups": int(rr.shape[0]), "cus_seen": len(per_cu), "launch_us": round((t1 - t0) / 100.0, 1),
                      "median_us": {"gap_between_workgroups_on_a_cu": us(gaps), "prologue_until_loop": us(pro), "chunk_loop": us(loop),
                                    "epilogue_until_stores_issued": us(epi), "store_drain": us(drain)}}), flush=True)


run("96 -> 96, 3x3x3, fused RMS-norm + SiLU, residual (slab kernel)", 21, 256, 256, 96, 96, True, True)
run("96 -> 96, 3x3x3, plain (slab kernel)", 21, 256, 256, 96, 96, False, False)
run("192 -> 192, 3x3x3, fused norm (prefetch kernel)", 21, 128, 128, 192, 192, True, True)
run("192 -> 192, 3x3x3, plain + residual (prefetch kernel)", 21, 128, 128, 192, 192, False, True)
run("384 -> 384, 3x3x3, plain (prefetch kernel, 2 N tiles)", 11, 64, 64, 384, 384, False, False)
run("96 -> 3 (+1 pad), 3x3x3, conv_out (slab kernel, narrow epilogue)", 21, 256, 256, 96, 3, False, False)
