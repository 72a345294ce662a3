#!/usr/bin/env python
"""Launches with fewer than 256 workgroups of 256 query rows (the 512^2 Flux geometry: 24 heads x 1536 = 144): the automatic rule
gives them to the 4-wave kernel (128-row workgroups, 288 of them); `attn.waves` = 8 forces the w64 kernel on 144 CUs.  Which wins?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
for H, S in [(24, 1536), (24, 2048), (24, 2560), (16, 1985), (40, 512), (24, 1024), (32, 2048)]:
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.zeros(1, H, 128, skp, device=DEV, dtype=torch.bfloat16)
    vt[..., :S] = torch.randn(1, H, 128, S, generator=g, device=DEV).to(torch.bfloat16)
    res, outs = {}, {}
    for rnd in range(3):
        for w in (0, 8):
            lib.tune_set("attn.waves", w)
            o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
            ops.attention_prepared(q, k, vt, o, S)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                ops.attention_prepared(q, k, vt, o, S)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / 50)
            outs[w] = o
    lib.tune_set("attn.waves", 0)
    nwg = ((S + 255) // 256) * H
    print(json.dumps({"H": H, "S": S, "workgroups_of_256_rows": nwg, "us": {w: round(min(v) * 1e3, 2) for w, v in res.items()},
                      "w64_over_auto": round(min(res[0]) / min(res[8]), 3),
                      "frac_differing": float((outs[0] != outs[8]).float().mean())}), flush=True)
