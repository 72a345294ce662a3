#!/usr/bin/env python
"""The 140..255-workgroup band of the small-launch rule with SHORT key sequences (cross-attention-like): w64 (rule, `attn.waves` 0)
against the 4-wave kernel (`attn.waves` 4), through the registered operator (V^T / packing passes included on both sides)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
for H, Sq, Sk in [(40, 1024, 64), (40, 1024, 128), (40, 1024, 256), (24, 1536, 512), (24, 2048, 77), (28, 1536, 1024), (24, 1536, 1536)]:
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, Sq, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, Sk, 128, generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn(1, H, Sk, 128, generator=g, device=DEV).to(torch.bfloat16)
    res = {}
    for rnd in range(3):
        for w in (0, 4):
            lib.tune_set("attn.waves", w)
            ops.attention(q, k, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                ops.attention(q, k, v)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / 50)
    lib.tune_set("attn.waves", 0)
    print(json.dumps({"H": H, "Sq": Sq, "Sk": Sk, "workgroups_of_256_rows": ((Sq + 255) // 256) * H,
                      "us": {("rule (w64)" if w == 0 else "4-wave"): round(min(x) * 1e3, 2) for w, x in res.items()},
                      "four_wave_over_rule": round(min(res[4]) / min(res[0]), 3)}), flush=True)
