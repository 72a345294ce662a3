#!/usr/bin/env python
"""Time the VAE decodes alone (run under rocprofv3 --kernel-trace --stats for the per-kernel split).
usage: vae_bench.py flux|wan|wan-untiled|wan-tile|hunyuan|taehv [reps]   (wan = the 4 x 7-tile decode the reference always
takes; wan-tile = ONE of its 28 tiles, [1,16,21,32,32] -> [1,3,81,256,256]: same launches, short enough for --pmc passes;
taehv = HunyuanVideo-1.5's light VAE on the same 480p x 121-frame latents as `hunyuan`)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from bench import synth_vae_init  # noqa: E402

for kv in filter(None, os.environ.get("TUNE", "").split(",")):        # TUNE="conv.v2=2" etc. (apexmi_tune_set keys)
    from apex_studio_amd import lib as _lib
    _lib.tune_set(kv.split("=")[0], int(kv.split("=")[1]))
which = sys.argv[1] if len(sys.argv) > 1 else "flux"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
if which == "flux":
    from apex_studio_amd.vae_flux import AutoencoderKL
    vae = synth_vae_init(AutoencoderKL(device=dev, dtype=torch.bfloat16), 5)
    z = torch.randn(1, 16, 128, 128, device=dev).to(torch.bfloat16)
elif which == "hunyuan":      # HunyuanVideo-1.5 480p x 121 frames: latent [32, 31, 30, 52], 8x8-latent tiles
    from apex_studio_amd.vae_hunyuan15 import AutoencoderKLHunyuanVideo15
    vae = synth_vae_init(AutoencoderKLHunyuanVideo15(device=dev, dtype=torch.bfloat16), 7)
    vae.enable_tiling()
    z = torch.randn(1, 32, 31, 30, 52, device=dev).to(torch.bfloat16)
elif which == "taehv":
    from apex_studio_amd.vae_taehv import AutoencoderKLHunyuanVideo15Light
    vae = synth_vae_init(AutoencoderKLHunyuanVideo15Light(device=dev), 8)
    z = torch.randn(1, 32, 31, 30, 52, device=dev).to(torch.bfloat16)
    _dec = vae.decode
    vae.decode = lambda zz, return_dict=False: _dec(zz)           # returns [1, N, 3, T', H', W']; [0] below = the video
else:
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    vae = synth_vae_init(AutoencoderKLWan(device=dev, dtype=torch.bfloat16), 6)
    if which != "wan-untiled":
        vae.enable_tiling()
    z = torch.randn(1, 16, 21, 90, 160, device=dev).to(torch.bfloat16)
    if which == "wan-tile":
        z = z[:, :, :, :32, :32].contiguous()
vae.decode(z, return_dict=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize()
print(f"{which} vae decode: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms  out {tuple(out.shape)}")
