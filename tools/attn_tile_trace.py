#!/usr/bin/env python
"""Per-workgroup timeline of the shipped flash-attention kernel at the Flux shape (24 heads x 4608 x 4608 x 128; 432 workgroups) and at
a Wan-like one, from a side library built with -DAPEXMI_ATTN_TRACE=1 (bash tools/gemm_tile_trace.sh build):
    APEX_MI355_LIB=tools/ubench/bin/libapex_trace.so python tools/attn_tile_trace.py"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
g = torch.Generator(device=DEV).manual_seed(0)
for name, H, S in (("flux 24 x 4608", 24, 4608), ("qwen-like 24 x 8448", 24, 8448), ("long 8 x 32768", 8, 32768)):
    skp = (S + 63) // 64 * 64
    q, k = (torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
    n = ((S + 255) // 256) * H
    tr = torch.zeros(n * 8, dtype=torch.int64, device=DEV)
    ops.attention_prepared(q, k, vt, out, S)
    torch.cuda.synchronize()
    os.environ["APEXMI_ATTN_TRACE_PTR"] = hex(tr.data_ptr())
    ops.attention_prepared(q, k, vt, out, S)
    torch.cuda.synchronize()
    os.environ.pop("APEXMI_ATTN_TRACE_PTR")
    r = tr.view(n, 8).cpu()
    r = r[r[:, 2] != 0]                      # workgroups of the main launch (a split tail writes no record)
    key = ((r[:, 1] & 0xf) << 16) | (r[:, 0] & 0xff00)
    per_cu = {}
    for i in range(r.shape[0]):
        per_cu.setdefault(int(key[i]), []).append([int(v) for v in r[i, 2:7]])
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
    print(json.dumps({"attention": name, "workgroups_recorded": int(r.shape[0]), "of": n, "cus_seen": len(per_cu),
                      "launch_us": round((t1 - t0) / 100.0, 1),
                      "median_us": {"gap_between_workgroups_on_a_cu": us(gaps), "prologue_until_first_tile_visible": us(pro),
                                    "kv_loop": us(loop), "epilogue_until_stores_issued": us(epi), "store_drain": us(drain)},
                      "kv_tiles": (S + 63) // 64}), flush=True)
