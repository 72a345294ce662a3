#!/usr/bin/env python
"""VERDICT r3 item 1d, the cheap decisive measurement BEFORE building it: does the MLP half of the single block's proj_out (K = 12288,
independent of attention) run underneath the attention launch on a side stream?  Per single block, production order is
    [QKV + MLP-up GEMM] -> attention (432 workgroups = 1.69 rounds) -> proj_out (K = 15360, 216 tiles).
The proposed order is  attention || proj_out[:, 3072:] (K = 12288, f32 partial)  ->  proj_out[:, :3072] (K = 3072) + partial.
This probe times, interleaved on one box, for 38 blocks with distinct (cold) weights:
    A  attention ; GEMM K = 15360                                  (production)
    B  attention ; GEMM K = 12288 ; GEMM K = 3072                  (the split alone, one stream: what the split costs)
    C  attention || GEMM K = 12288 (side stream) ; GEMM K = 3072   (the overlap, without the partial's extra 113 MB of traffic)
    D  as C with the attention enqueued first;  E  attention on a high-priority stream
so the best of C / D / E is an UPPER bound of what item 1d can give.  Prints ms per 38 blocks and the bound in ms / step."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402

DEV = "cuda"
S, H, D, DIM, MLP = 4608, 24, 128, 3072, 12288
NB = 38
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV)  # noqa: E731
W = [(rn(DIM, DIM + MLP) * (DIM + MLP) ** -0.5).to(torch.bfloat16) for _ in range(NB)]
CAT = rn(S, DIM + MLP).to(torch.bfloat16)
X = torch.empty(S, DIM, device=DEV, dtype=torch.bfloat16)
P = torch.empty(S, DIM, device=DEV, dtype=torch.bfloat16)
Skp = (S + 63) // 64 * 64
Q, K = rn(1, H, S, D).to(torch.bfloat16), rn(1, H, S, D).to(torch.bfloat16)
VT = rn(1, H, D, Skp).to(torch.bfloat16)
att_v = CAT[:, :DIM].view(1, S, H, D)
side = torch.cuda.Stream(device=DEV)


def arm_a():
    for w in W:
        ops.attention_prepared(Q, K, VT, att_v, S)
        ops.gemm(CAT, w, None, out=X)


def arm_b():
    for w in W:
        ops.attention_prepared(Q, K, VT, att_v, S)
        ops.gemm(CAT[:, DIM:], w[:, DIM:], None, out=P)
        ops.gemm(CAT[:, :DIM], w[:, :DIM], None, out=X)


def arm_c():
    main = torch.cuda.current_stream()
    for w in W:
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm(CAT[:, DIM:], w[:, DIM:], None, out=P)
            done = torch.cuda.Event()
            done.record(side)
        ops.attention_prepared(Q, K, VT, att_v, S)
        main.wait_event(done)
        ops.gemm(CAT[:, :DIM], w[:, :DIM], None, out=X)


def arm_d():      # as C, attention enqueued FIRST (the dispatcher sees its workgroups before the GEMM's)
    main = torch.cuda.current_stream()
    for w in W:
        ev = torch.cuda.Event()
        ev.record(main)
        ops.attention_prepared(Q, K, VT, att_v, S)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm(CAT[:, DIM:], w[:, DIM:], None, out=P)
            done = torch.cuda.Event()
            done.record(side)
        main.wait_event(done)
        ops.gemm(CAT[:, :DIM], w[:, :DIM], None, out=X)


low = torch.cuda.Stream(device=DEV, priority=0)
high = torch.cuda.Stream(device=DEV, priority=-1)


def arm_e():      # attention on a high-priority stream, the GEMM half on a normal one
    cur = torch.cuda.current_stream()
    high.wait_stream(cur)
    for w in W:
        ev = torch.cuda.Event()
        ev.record(high)
        with torch.cuda.stream(low):
            low.wait_event(ev)
            ops.gemm(CAT[:, DIM:], w[:, DIM:], None, out=P)
            done = torch.cuda.Event()
            done.record(low)
        with torch.cuda.stream(high):
            ops.attention_prepared(Q, K, VT, att_v, S)
            high.wait_event(done)
            ops.gemm(CAT[:, :DIM], w[:, :DIM], None, out=X)
    cur.wait_stream(high)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


arms = {"A_production": arm_a, "B_split_one_stream": arm_b, "C_split_side_stream": arm_c, "D_attention_enqueued_first": arm_d,
        "E_attention_high_priority": arm_e}
for fn in arms.values():
    fn()
res = {k: [] for k in arms}
for r in range(int(os.environ.get("ROUNDS", "7"))):
    for k, fn in arms.items():
        res[k].append(timed(fn))
med = {k: statistics.median(v) for k, v in res.items()}
print(json.dumps({"ms_per_38_blocks_median": med, "min": {k: min(v) for k, v in res.items()},
                  "upper_bound_gain_ms_per_step": med["A_production"] - min(med[k] for k in med if k[0] in "CDE"),
                  "split_cost_ms_per_step": med["B_split_one_stream"] - med["A_production"]}))
