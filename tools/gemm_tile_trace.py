#!/usr/bin/env python
"""Per-workgroup timeline of the shipped GEMM schedule: where a tile's fixed cost (tools/gemm_k_sweep.py: 13-18 us per tile round)
goes.  Needs the side library built with -DAPEXMI_GEMM_TRACE=1 (bash tools/gemm_tile_trace.sh build):
    APEX_MI355_LIB=tools/ubench/bin/libapex_trace.so python tools/gemm_tile_trace.py
Every workgroup's wave 0 records s_memrealtime (10 ns ticks) at kernel entry, K-loop begin, K-loop end, after the last epilogue store
was ISSUED and after all of them were ACKNOWLEDGED (an extra vmcnt(0) the shipped kernel does not execute), plus HW_ID / XCC_ID.  Per CU
(xcc, se, sh, cu) the workgroups are sorted by entry time:
    gap       = entry of a workgroup - acknowledged time of its predecessor on the same CU (hand-over; negative = the hardware started
                it before the predecessor's stores were acknowledged)
    prologue  = entry -> K-loop begin (addresses, first stage in LDS)
    loop      = K-loop
    epilogue  = K-loop end -> last store issued (bias / gate / residual loads, arithmetic, stores)
    drain     = last store issued -> acknowledged
Medians over the launch, in us."""
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
g = torch.Generator(device=DEV).manual_seed(0)
CASES = [("ff_up K=3072 gelu", 4608, 12288, 3072, "gelu"), ("attn_out K=3072 gate_res", 4608, 3072, 3072, "gate_res"),
         ("proj_out K=15360 gate_res", 4608, 3072, 15360, "gate_res"), ("ff_up K=256 bias", 4608, 12288, 256, "bias")]


def set_trace(t):
    p = t.data_ptr() if t is not None else 0
    lib.tune_set("gemm.trace_lo", ctypes.c_int32(p & 0xffffffff).value)
    lib.tune_set("gemm.trace_hi", ctypes.c_int32(p >> 32).value)


def report(name, tr, ntiles, sel=None):
    r = tr.view(ntiles, 8).cpu()
    hw, xcc = r[:, 0], r[:, 1]
    cu_key = ((xcc & 0xf) << 16) | (hw & 0xff00)                 # xcc | se, sh, cu of HW_ID[15:8]
    per_cu = {}
    for i in range(ntiles):
        per_cu.setdefault(int(cu_key[i]), []).append([int(v) for v in r[i, 2:7]] + [int(r[i, 7])])
    gaps, pro, loop, epi_t, drain, first_entry, last_ack = [], [], [], [], [], None, None
    for rows in per_cu.values():
        rows.sort()
        for j, (t_in, l0, l1, st, ack, seq) in enumerate(rows):
            if sel is None or sel(seq):
                pro.append(l0 - t_in)
                loop.append(l1 - l0)
                epi_t.append(st - l1)
                drain.append(ack - st)
            if j:
                gaps.append(t_in - rows[j - 1][4])
            first_entry = t_in if first_entry is None else min(first_entry, t_in)
            last_ack = ack if last_ack is None else max(last_ack, ack)
    us = lambda v: round(statistics.median(v) / 100.0, 2) if v else None  # noqa: E731
    p90 = lambda v: round(sorted(v)[int(0.9 * len(v))] / 100.0, 2) if v else None  # noqa: E731
    print(json.dumps({"gemm": name, "tiles": ntiles, "tiles_in_median": len(pro), "cus_seen": len(per_cu),
                      "launch_us_first_entry_to_last_ack": round((last_ack - first_entry) / 100.0, 1),
                      "median_us": {"gap_between_workgroups_on_a_cu": us(gaps), "prologue": us(pro), "k_loop": us(loop),
                                    "epilogue_until_stores_issued": us(epi_t), "store_drain": us(drain)},
                      "p90_us": {"gap": p90(gaps), "prologue": p90(pro), "epilogue": p90(epi_t), "drain": p90(drain)}}), flush=True)


# the single block's fused launch: QKV (q / k norm + RoPE + [H, S, D] / V^T layout in the epilogue) + MLP-up (gelu) reading the same XN
S, H, DIM, MLP = 4608, 24, 3072, 12288
xn = torch.randn(S, DIM, generator=g, device=DEV).to(torch.bfloat16)
wq = [(torch.randn(3 * DIM, DIM, generator=g, device=DEV) * DIM ** -0.5).to(torch.bfloat16) for _ in range(2)]
wm = [(torch.randn(MLP, DIM, generator=g, device=DEV) * DIM ** -0.5).to(torch.bfloat16) for _ in range(2)]
bq, bm = torch.randn(3 * DIM, generator=g, device=DEV).to(torch.bfloat16), torch.randn(MLP, generator=g, device=DEV).to(torch.bfloat16)
nq, nk = (1 + 0.1 * torch.randn(128, generator=g, device=DEV)).to(torch.bfloat16), (1 + 0.1 * torch.randn(128, generator=g, device=DEV)).to(torch.bfloat16)
rope = torch.randn(2, S, 128, generator=g, device=DEV)
skp = (S + 63) // 64 * 64
qo = torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16)
ko = torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16)
vt = torch.zeros(H, 128, skp, device=DEV, dtype=torch.bfloat16)
mlp_out = torch.empty(S, MLP, device=DEV, dtype=torch.bfloat16)


def fused(i):
    ops.gemm_grouped_qkv([xn, xn], [wq[i], wm[i]], [bq, bm], [None, mlp_out], ["bias", "gelu"], [1, 0], [nq, None], [nk, None], [0, 0],
                         H, 1e-6, rope, qo, ko, vt)


nt_q, nt_all = 18 * 36, 18 * 36 + 18 * 48
tr = torch.zeros(nt_all * 8, dtype=torch.int64, device=DEV)
fused(0)
torch.cuda.synchronize()
set_trace(tr)
fused(1)
torch.cuda.synchronize()
set_trace(None)
report("single-block fused launch: its QKV tiles (q/k norm + RoPE / V^T epilogue)", tr, nt_all, sel=lambda seq: seq < nt_q)
report("single-block fused launch: its MLP-up tiles (gelu epilogue)", tr, nt_all, sel=lambda seq: seq >= nt_q)

for name, M, N, K, epi in CASES:
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    ws = [(torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16) for _ in range(3)]
    bias = torch.randn(N, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    kw = {}
    if epi == "gate_res":
        kw = dict(gate=torch.randn(N, generator=g, device=DEV), residual=out)
    ntiles = ((M + 255) // 256) * ((N + 255) // 256)
    tr = torch.zeros(ntiles * 8, dtype=torch.int64, device=DEV)
    for w in ws[:2]:                       # warm: clocks up, the trace pointer still unset
        ops.gemm(a, w, bias, out=out, epilogue=epi, **kw)
    torch.cuda.synchronize()
    set_trace(tr)
    ops.gemm(a, ws[2], bias, out=out, epilogue=epi, **kw)      # cold weights
    torch.cuda.synchronize()
    set_trace(None)
    report(name, tr, ntiles)
