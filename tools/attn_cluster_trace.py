#!/usr/bin/env python
"""Cluster-level timeline of the shipped flash-attention loop (attn_fwd_d128_c4_kernel): how many core cycles each of the four
clusters of a KV tile takes in the early half (wave 0) and in the late half (wave 4) of a workgroup, split into the cluster's own
work (barrier exit -> its closing wait) and the wait for the closing barrier.  Side library built with -DAPEXMI_ATTN_TRACE=2
(bash tools/attn_cluster_trace.sh build):
    APEX_MI355_LIB=tools/ubench/bin/libapex_trace2.so python tools/attn_cluster_trace.py
Stamps are s_memtime (core clock) of tiles 16..23 of every workgroup; the stamping itself costs a cluster a few tens of cycles
(one SMEM issue at each end, one ds_write of the previous pair), so the numbers are for ranking the clusters, not for the third digit."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
lib.tune_set("attn.w64", 0)   # this tool measures the 4-cluster kernel (the shipped main launch is attn.w64 = 1)
NAMES = ["C1 K reads (+DMA)", "C2 QK^T", "C3 V reads + softmax", "C4 PV"]
ARMS = [tuple(int(x) for x in a.split(":")) for a in os.environ.get("ARMS", "2:0,2:4").split(",")]
g = torch.Generator(device=DEV).manual_seed(0)
for name, H, S in (("flux 24 x 4608", 24, 4608), ("long 8 x 32768", 8, 32768)):
    skp = (S + 63) // 64 * 64
    q, k = (torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(1, S, H, 128, device=DEV, dtype=torch.bfloat16)
    n = ((S + 255) // 256) * H
    for stages, dma in ARMS:
        lib.tune_set("attn.stages", stages)
        lib.tune_set("attn.xv", dma)
        tr = torch.zeros(n * 264, dtype=torch.int64, device=DEV)
        for _ in range(3):
            ops.attention_prepared(q, k, vt, out, S)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.attention_prepared(q, k, vt, out, S)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        os.environ["APEXMI_ATTN_TRACE_PTR"] = hex(tr.data_ptr())
        ops.attention_prepared(q, k, vt, out, S)
        torch.cuda.synchronize()
        os.environ.pop("APEXMI_ATTN_TRACE_PTR")
        r = tr.view(n, 264).cpu()
        r = r[r[:, 2] != 0]
        st = r[:, 8:].reshape(-1, 2, 8, 4, 4).double()      # [wg, half, tile, cluster, (start, end, mid, -)]
        work = st[..., 1] - st[..., 0]                       # cluster's own work
        head = st[..., 2] - st[..., 0]                       # ... of which up to the mid stamp (reads issued / last MFMA issued)
        head2 = st[..., 3] - st[..., 0]                      # C3 only: up to the row max known
        flat = st.reshape(-1, 2, 32, 4)                      # clusters in program order
        nxt = flat[:, :, 1:, 0] - flat[:, :, :-1, 1]         # closing wait = next start - this end
        wait = torch.cat([nxt, torch.full_like(nxt[:, :, :1], float("nan"))], dim=2).reshape(-1, 2, 8, 4)
        tile = (flat[:, :, 4:, 0] - flat[:, :, :-4, 0]).reshape(-1, 2, 28)   # start-to-start over four clusters
        rec = {"attention": name, "stages": stages, "xv": dma, "ms": round(ms, 4),
               "tflops": round(4.0 * H * S * S * 128 / (ms * 1e-3) / 1e12, 1), "workgroups": int(r.shape[0]),
               "cycles_per_tile_median": [round(float(tile[:, h].median()), 0) for h in (0, 1)]}
        for h, hn in ((0, "early"), (1, "late")):
            rec[hn] = {NAMES[c]: {"work": round(float(work[:, h, :, c].median()), 0),
                                  "to_mid": round(float(head[:, h, :, c].median()), 0),
                                  **({"to_max": round(float(head2[:, h, :, c].median()), 0)} if c == 2 else {}),
                                  "wait": round(float(wait[:, h, :, c].nanmedian()), 0)} for c in range(4)}
        print(json.dumps(rec), flush=True)
lib.tune_set("attn.stages", 2)
lib.tune_set("attn.xv", 0)
