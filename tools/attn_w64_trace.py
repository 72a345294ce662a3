#!/usr/bin/env python
"""Core cycles per KV tile of the w64 attention loop, by phase (side library of tools/attn_w64_ablate.sh, attn.w64 = 2 = the traced
body): phase X (S(t+1) MFMAs + exp / sum / bf16), phase Y (P V MFMAs + max / decision / scaling), wait for the LDS-DMA, barrier +
scalar bookkeeping.  Cycle counts do not depend on the clock the board picks, wall time does.
    APEX_MI355_LIB=tools/ubench/bin/libapex_w64abl.so python tools/attn_w64_trace.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
VARS = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "bin", "w64abl_variants.txt")).read().strip()
print(VARS, flush=True)
for name, H, S in (("long", 8, 32768),):
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q, k = (torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
    n = ((S + 255) // 256) * H
    for var in [int(x) for x in os.environ.get("VARIANTS", "2,3,4,5,6,7").split(",")]:
        tr = torch.zeros(n * 16, dtype=torch.int64, device=DEV)
        lib.tune_set("attn.w64", var)
        ops.attention_prepared(q, k, vt, o, S)
        torch.cuda.synchronize()
        os.environ["APEXMI_ATTN_TRACE_PTR"] = hex(tr.data_ptr())
        ops.attention_prepared(q, k, vt, o, S)
        torch.cuda.synchronize()
        os.environ.pop("APEXMI_ATTN_TRACE_PTR")
        lib.tune_set("attn.w64", 0)
        r = tr.view(n, 4, 4).double().cpu() / (skp // 64)          # [workgroup, wave, (X, Y, dma wait, barrier)] cycles per tile
        med = r.median(dim=0).values.mean(dim=0)
        print(json.dumps({"shape": name, "variant": var, "cycles_per_tile": {"X": round(float(med[0])), "Y": round(float(med[1])),
                          "dma_wait": round(float(med[2])), "barrier": round(float(med[3])), "sum": round(float(med.sum()))}}), flush=True)
