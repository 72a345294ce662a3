#!/usr/bin/env python
"""Timing of the w64 attention loop with parts left out (side library of tools/attn_w64_ablate.sh; results are WRONG by design):
    APEX_MI355_LIB=tools/ubench/bin/libapex_w64abl.so python tools/attn_w64_ablate.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
NAMES = {0: "shipped 4-cluster", 1: "w64 full", 2: "w64 no X fillers", 3: "w64 no Y fillers", 4: "w64 no fillers",
         5: "w64 no fillers no drain", 6: "w64 no MFMAs", 7: "w64 no fillers no DMA"}
ARMS = [int(x) for x in os.environ.get("ARMS", "0,1,2,3,4,5,6,7").split(",")]
for name, H, S in (("flux", 24, 4608), ("long", 8, 32768)):
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q, k = (torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
    iters = 30 if S < 20000 else 3
    res = {}
    for rnd in range(2):
        for w in ARMS:
            lib.tune_set("attn.w64", w)
            ops.attention_prepared(q, k, vt, o, S)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                ops.attention_prepared(q, k, vt, o, S)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / iters)
    lib.tune_set("attn.w64", 0)
    nt = skp // 64
    rounds = ((S + 255) // 256) * H / 256.0
    print(json.dumps({"shape": name, "ms": {NAMES[w]: round(min(v), 4) for w, v in res.items()},
                      "us_per_tile_at_full_rounds": {NAMES[w]: round(min(v) * 1e3 / (nt * max(rounds, 1.0)), 3) for w, v in res.items()}}),
          flush=True)
