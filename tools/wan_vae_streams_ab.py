#!/usr/bin/env python
"""Wan 720p x 81-frame tiled VAE decode with the tiles on 1 / 2 / 3 / 4 HIP streams: time and bit-identity."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: F401
from bench import synth_vae_init
from apex_studio_amd.vae_wan import AutoencoderKLWan
dev = torch.device("cuda", 0)
vae = synth_vae_init(AutoencoderKLWan(device=dev, dtype=torch.bfloat16), 5)
vae.enable_tiling()
z = torch.randn(1, 16, 21, 90, 160, device=dev).to(torch.bfloat16)
ref = None
for ns in (1, 2, 1, 2, 3, 4):
    vae.tile_streams = ns
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if ref is None:
        ref = o
    print(f"tile_streams={ns}: {dt*1e3:.1f} ms  equal_to_sequential={torch.equal(o, ref)}  mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
