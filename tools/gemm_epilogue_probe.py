#!/usr/bin/env python
"""Where the Flux single block's QKV + MLP-up launch (4608 x 21504 x 3072; 443 us with plain epilogues, 526 us in the step) spends
the difference: the same launch, cold weights (38 layers in sequence), with
  plain        one problem, bias epilogue                                    (384 x 256 tiling)
  gelu         two problems [9216 bias | 12288 GELU]                         (384 x 256)
  qkv          two problems [9216 fused q/k/v | 12288 GELU] = the step's     (384 x 256, `gemm.x384_qkv` 1: shipped)
  qkv256       the same on the 256 x 256 tiling                              (`gemm.x384_qkv` 0)
  split        q/k/v-fused 9216 on 256 x 256 + GELU 12288 as its own launch  (two launches)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib, ops  # noqa: E402

DEV = "cuda"
L, S, K, H = 38, 4608, 3072, 24
dim, mlp = 3072, 12288
g = torch.Generator(device=DEV).manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(torch.bfloat16)


a = rnd(S, K)
wq = [rnd(3 * dim, K, scale=K ** -0.5) for _ in range(L)]
wm = [rnd(mlp, K, scale=K ** -0.5) for _ in range(L)]
wcat = [torch.cat([q, m], 0) for q, m in zip(wq, wm)]
bq, bm = rnd(3 * dim, scale=0.1), rnd(mlp, scale=0.1)
bcat = torch.cat([bq, bm])
nq, nk = rnd(128) * 0.2 + 1, rnd(128) * 0.2 + 1
ang = torch.rand(S, 64, generator=g, device=DEV) * 6.283
rope = torch.stack([ang.cos().repeat_interleave(2, 1), ang.sin().repeat_interleave(2, 1)]).contiguous().float()
out_all = torch.empty(S, 3 * dim + mlp, device=DEV, dtype=torch.bfloat16)
qkv, cat = torch.empty(S, 3 * dim, device=DEV, dtype=torch.bfloat16), torch.empty(S, mlp, device=DEV, dtype=torch.bfloat16)
Q, Kk = (torch.empty(H, S, 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
VT = torch.zeros(H, 128, (S + 63) // 64 * 64, device=DEV, dtype=torch.bfloat16)


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / L * 1e3


def fused(l):
    ops.gemm_grouped_qkv([a, a], [wq[l], wm[l]], [bq, bm], [None, cat], ["bias", "gelu"], [1, 0], [nq, None], [nk, None], [0, 0], H,
                         1e-6, rope, Q, Kk, VT)


arms = {
    "plain": lambda: [ops.gemm(a, wcat[l], bcat, out=out_all) for l in range(L)],
    "gelu": lambda: [ops.gemm_grouped([a, a], [wq[l], wm[l]], [bq, bm], [qkv, cat], epilogue=["bias", "gelu"]) for l in range(L)],
    "qkv": lambda: [fused(l) for l in range(L)],
    "split": lambda: [(ops.gemm_grouped_qkv([a], [wq[l]], [bq], [None], "bias", [1], [nq], [nk], [0], H, 1e-6, rope, Q, Kk, VT),
                       ops.gemm(a, wm[l], bm, out=cat, epilogue="gelu")) for l in range(L)],
}
# the double block's grouped launch (image 4096 + text 512 rows, N 9216) on the 256 x 256 tiling: plain vs fused q/k/v
ai, at = rnd(4096, K), rnd(512, K)
wq2 = [rnd(3 * dim, K, scale=K ** -0.5) for _ in range(L)]
qkv_i, qkv_t = torch.empty(4096, 3 * dim, device=DEV, dtype=torch.bfloat16), torch.empty(512, 3 * dim, device=DEV, dtype=torch.bfloat16)
arms["dbl_plain"] = lambda: [ops.gemm_grouped([ai, at], [wq[l], wq2[l]], [bq, bq], [qkv_i, qkv_t]) for l in range(L)]
arms["dbl_qkv"] = lambda: [ops.gemm_grouped_qkv([ai, at], [wq[l], wq2[l]], [bq, bq], [None, None], "bias", [1, 1], [nq, nq], [nk, nk], [512, 0],
                                                H, 1e-6, rope, Q, Kk, VT) for l in range(L)]
res = {}
only = os.environ.get("ARMS")
if only:
    arms = {k: v for k, v in arms.items() if k in only.split(",")}
for rnd_ in range(2):
    for name, fn in arms.items():
        res.setdefault(name, []).append(round(timeit(fn), 1))
    if not only:
        lib.tune_set("gemm.x384_qkv", 0)
        res.setdefault("qkv256", []).append(round(timeit(arms["qkv"]), 1))
        lib.tune_set("gemm.x384_qkv", 1)
res["pairs_off"] = []
ops.rope_pairs_enabled = False        # the direct-load epilogue (full table), same binary
for rnd_ in range(2):
    res["pairs_off"].append({k: round(timeit(arms[k]), 1) for k in ("qkv", "dbl_qkv") if k in arms})
ops.rope_pairs_enabled = True
print(json.dumps({"us_per_launch": res, "best": {k: min(v) for k, v in res.items() if k != "pairs_off"}}))
