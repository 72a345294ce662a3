#!/usr/bin/env python
"""Where the per-clip set-up of a Flux clip (FluxTransformer2DModel.begin_schedule: the modulation table of n steps) spends its time:
wall time with a sync after each part, first call and repeats, for n = 5 / 20 / 28 rows.  python tools/begin_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
from apex_studio_amd.flux import FluxTransformer2DModel  # noqa: E402
model = FluxTransformer2DModel(**dict(bench.FLUX_DEV), device=dev, dtype=torch.bfloat16).init_synthetic(seed=1234)
model.pack()
gs = torch.Generator(device=dev).manual_seed(7)
pooled = torch.randn(1, 768, generator=gs, device=dev).to(torch.bfloat16)
guidance = torch.full([1], 3.5, device=dev, dtype=torch.float32)


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


for n in (5, 20, 20, 28, 28, 20):
    ts = torch.linspace(1.0, 0.05, n, device=dev)
    t_all, h = wall(lambda: model.begin_schedule(ts, guidance, pooled))
    t_end, _ = wall(lambda: model.end_schedule(h))
    # the parts, by hand
    B = 1
    tt = ts.reshape(n, 1).expand(n, B)
    g = guidance.reshape(1, -1).expand(n, B)
    t_cond, cond = wall(lambda: model._cond_rows(tt.reshape(-1), g.reshape(-1), pooled.unsqueeze(0).expand(n, B, 768).reshape(n * B, -1)))
    t_gemv, tab = wall(lambda: ops.gemv(model._mod_w, cond, model._mod_b, pre_silu=True))
    del tab
    print(f"n={n:2d}: begin_schedule {t_all:7.2f} ms  end_schedule {t_end:6.2f} ms | cond rows {t_cond:6.2f} ms  table gemv {t_gemv:6.2f} ms "
          f"(weights {model._mod_w.numel() * 2 / 1e9:.2f} GB)", flush=True)
