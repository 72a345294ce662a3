#!/usr/bin/env python
"""Variants of the generated w64 attention loop (side library of tools/attn_w64_variants.sh) against the shipped loop: rel-L2 of
every arm vs the float64 softmax attention on small / ragged shapes, then interleaved timing on the BASELINE geometries.
    APEX_MI355_LIB=tools/ubench/bin/libapex_w64var.so ARMS=1,2,3 SHAPES=flux,long,wan python tools/attn_w64_variants.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
ARMS = [int(x) for x in os.environ.get("ARMS", "1,2").split(",")]


def ref_sdpa(q, k, v):
    s = (q.double() @ k.double().transpose(-1, -2)) * q.shape[-1] ** -0.5
    return torch.softmax(s, -1) @ v.double()


lib.tune_set("attn.waves", 8)
try:
    for B, H, Sq, Sk in [(1, 2, 256, 64), (1, 2, 256, 192), (1, 2, 700, 333), (2, 3, 260, 1000), (1, 4, 1536, 1536), (1, 3, 513, 4097),
                         (1, 2, 1024, 16384)]:
        g = torch.Generator(device=DEV).manual_seed(Sq * 7 + Sk)
        q, k, v = (torch.randn(B, H, s_, 128, generator=g, device=DEV).to(torch.bfloat16) for s_ in (Sq, Sk, Sk))
        if Sk == 16384:              # peaked rows: one key towers late in the sequence (the rescale path), a few rows nearly uniform
            k[:, :, 9000] = q[:, :, 5:6].mean(2) * 6
            q[:, :, :64] *= 0.02
        ref = ref_sdpa(q, k, v)
        row = {"shape": [B, H, Sq, Sk]}
        for w in ARMS:
            lib.tune_set("attn.w64", w)
            o = ops.attention(q, k, v)
            assert torch.equal(o, ops.attention(q, k, v)), f"arm {w} not deterministic"
            o = o.double()
            o = o if o.shape == ref.shape else o.transpose(1, 2)
            row[f"rel_l2_vs_f64[{w}]"] = float((o - ref).norm() / ref.norm())
            row[f"rel_l2_bf16_of_f64[{w}]"] = float((o - ref.to(torch.bfloat16).double()).norm() / ref.norm())
        print(json.dumps(row), flush=True)
finally:
    lib.tune_set("attn.waves", 0)
    lib.tune_set("attn.w64", 1)

SHAPES = {"flux": (24, 4608), "qwen": (24, 8448), "long": (8, 32768), "wan": (40, 75600)}
for name in os.environ.get("SHAPES", "flux,long,wan").split(","):
    H, S = SHAPES[name]
    skp = (S + 63) // 64 * 64
    g = torch.Generator(device=DEV).manual_seed(H)
    q = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    k = torch.randn(1, H, S, 128, generator=g, device=DEV).to(torch.bfloat16)
    vt = torch.randn(1, H, 128, skp, generator=g, device=DEV).to(torch.bfloat16)
    o = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=DEV)
    iters = 40 if S < 20000 else 3
    best = {}
    for rnd in range(3):
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
            best.setdefault(w, []).append(e0.elapsed_time(e1) / iters)
    lib.tune_set("attn.w64", 1)
    print(json.dumps({"shape": name, "H": H, "S": S, "ms": {w: [round(x, 4) for x in best[w]] for w in ARMS},
                      "tflops": {w: round(4.0 * H * S * S * 128 / (min(best[w]) * 1e-3) / 1e12, 1) for w in ARMS}}), flush=True)
