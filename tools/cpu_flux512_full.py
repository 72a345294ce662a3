#!/usr/bin/env python
"""BASELINE.json configs[0] as the reference itself runs it: Flux-Dev 512 x 512, 4 steps, the CPU PyTorch path — timed IN FULL
on the GPU box's host cores (bench.py's `cpu_baseline` extrapolates two blocks, the whole
4-step clip had never been timed).  The oracle (oracle/flux.py, fp32, the CPU restatement of the reference's
FluxTransformer2DModel pinned by tests/golden/flux_hybrid.pt) at full depth (19 + 38 blocks, d 3072, S 1024 + 512), random-init
weights, 4 FlowMatch-Euler steps, no VAE.  STEPS=4 THREADS=<all>.  Prints one JSON record (copy to profiles/)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import best_threads, cpu_info  # noqa: E402
from oracle import flux as OF  # noqa: E402
from oracle import layers as OL  # noqa: E402

STEPS = int(os.environ.get("STEPS", "4"))
sweep = None
if os.environ.get("THREADS"):
    threads = int(os.environ["THREADS"])
else:           # the thread count one full-width single block at this sequence runs fastest at (all logical CPUs is not it)
    _g = torch.Generator().manual_seed(1)
    _blk = OF.FluxSingleTransformerBlock(3072, 24, 128).eval()
    _x, _c, _t = torch.randn(1, 1024, 3072, generator=_g), torch.randn(1, 512, 3072, generator=_g), torch.randn(1, 3072, generator=_g)
    _rope = OF.flux_pos_embed(torch.cat((torch.zeros(512, 3), OF.latent_image_ids(32, 32)), dim=0), (16, 56, 56))
    with torch.no_grad():
        threads, sweep = best_threads(lambda: _blk(_x, _c, _t, _rope, OL.FP32), budget_s=60.0)
    del _blk, _x, _c, _t
torch.set_num_threads(threads)
LD, LS = (int(v) for v in os.environ.get("LAYERS", "19,38").split(","))        # debug only: anything but 19,38 is not the model
cfg = dict(patch_size=1, in_channels=64, num_layers=LD, num_single_layers=LS, attention_head_dim=128, num_attention_heads=24,
           joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
t0 = time.perf_counter()
with torch.no_grad():
    m = OF.FluxTransformer2DModel(**cfg).eval()
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.01))
t_init = time.perf_counter() - t0
lat = torch.randn(1, 1024, 64, generator=g)
enc, pooled = torch.randn(1, 512, 4096, generator=g), torch.randn(1, 768, generator=g)
img_ids, txt_ids, guid = OF.latent_image_ids(32, 32), torch.zeros(512, 3), torch.full([1], 3.5)
sig = torch.linspace(1.0, 1.0 / STEPS, STEPS).tolist() + [0.0]
per = []
with torch.no_grad():
    for i in range(STEPS):
        t1 = time.perf_counter()
        v = m(lat, enc, pooled, torch.tensor([sig[i]]), img_ids, txt_ids, guid, policy=OL.FP32)
        lat = lat + (sig[i + 1] - sig[i]) * v
        per.append(time.perf_counter() - t1)
        print(f"step {i}: {per[-1]:.1f} s", file=sys.stderr, flush=True)
print(json.dumps({"what": "Flux-Dev 512x512, 4 steps, CPU PyTorch fp32 (oracle = CPU restatement of the reference path), full depth 19 + 38 "
                          "blocks, S 1024 + 512, B = 1, no CFG, Euler update, no VAE", "steps": STEPS, "seconds_per_step": per,
                  "layers": [LD, LS], "clip_seconds": sum(per), "steps_per_sec": STEPS / sum(per), "threads": threads, "thread_sweep_single_block_s": sweep, "init_seconds": t_init,
                  "finite": bool(torch.isfinite(lat).all()), **cpu_info()}))
