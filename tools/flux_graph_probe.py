#!/usr/bin/env python
"""Does a HIP graph shorten the Flux step?  One denoise step (fixed modulation row, static buffers) captured with torch.cuda.CUDAGraph and
replayed, against the same step launched eagerly: ms per step, interleaved.  An upper bound of what graph launch could give the engine
(the real loop would need one graph per step or a device-side step index)."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd.engine_flux import calculate_shift, latent_image_ids  # noqa: E402
from apex_studio_amd.flux import FluxTransformer2DModel  # noqa: E402
from tools.flux_step_ab import FLUX_DEV  # noqa: E402

DEV = "cuda"
model = FluxTransformer2DModel(**FLUX_DEV, device=DEV, dtype=torch.bfloat16).init_synthetic(seed=1234)
model.pack()
g = torch.Generator(device=DEV).manual_seed(100)
lat = torch.randn(1, 4096, 64, generator=g, device=DEV).to(torch.bfloat16)
enc = torch.randn(1, 512, 4096, generator=g, device=DEV).to(torch.bfloat16)
pooled = torch.randn(1, 768, generator=g, device=DEV).to(torch.bfloat16)
img_ids, txt_ids = latent_image_ids(64, 64).to(DEV), torch.zeros(512, 3, device=DEV)
guidance = torch.full([1], 3.5, device=DEV, dtype=torch.float32)
ts = torch.linspace(1.0, 1.0 / 12, 12, device=DEV).to(torch.bfloat16)
model.begin_schedule(torch.stack([t.expand(1) for t in ts]), guidance, pooled)
out = torch.empty_like(lat)


def step():
    v = model(hidden_states=lat, timestep=ts[3].expand(1), guidance=guidance, pooled_projections=pooled, encoder_hidden_states=enc,
              txt_ids=txt_ids, img_ids=img_ids, joint_attention_kwargs={"modulation_step": 3}, return_dict=False)[0]
    out.copy_(v)


for _ in range(3):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref = out.clone()
graph.replay()
torch.cuda.synchronize()
same = bool(torch.equal(ref, out))


def timed(fn, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {"eager": [], "graph": []}
for r in range(5):
    res["eager"].append(timed(step))
    res["graph"].append(timed(graph.replay))
print(json.dumps({"ms_per_step_median": {k: round(statistics.median(v), 3) for k, v in res.items()},
                  "all": {k: [round(x, 2) for x in v] for k, v in res.items()}, "replay_bit_identical": same}))
