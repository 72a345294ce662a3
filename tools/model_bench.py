#!/usr/bin/env python
"""Full-size single-GPU timings of the other BASELINE configs (synthetic weights/inputs):
  qwen  — QwenImage-Edit-2509 1024^2 + one 1024^2 condition image, one denoise step (config 3)
  wan   — Wan-2.2 A14B 720p x 81f, one expert forward (config 4), optional
  vae   — Wan 3D-VAE decode 21x90x160 -> 81x720x1280 with the reference's 4x7 tiling (config 4)
Prints one JSON line per workload."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib  # noqa: E402

DEV = torch.device("cuda", 0)


def timed(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def kernels(fn):
    lib.prof_reset()
    lib.prof_enable(True)
    fn()
    p = lib.prof_read()
    lib.prof_enable(False)
    lib.prof_reset()
    return {k: {"ms": round(v["ms"], 2), "launches": v["launches"],
                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None}
            for k, v in p.items() if v["launches"]}


def qwen():
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    m = QwenImageTransformer2DModel(device=DEV, dtype=torch.bfloat16).init_synthetic(1)
    m.pack()
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(1, 8192, 64, generator=g, device=DEV).to(torch.bfloat16)
    enc = torch.randn(1, 256, 3584, generator=g, device=DEV).to(torch.bfloat16)
    t = torch.tensor([0.5], device=DEV)
    shapes = [[(1, 64, 64), (1, 64, 64)]]
    fn = lambda: m(hidden_states=x, encoder_hidden_states=enc, encoder_hidden_states_mask=None, timestep=t,  # noqa: E731
                   img_shapes=shapes, txt_seq_lens=[256], return_dict=False)[0]
    s = timed(fn, n=5, warm=2)
    print(json.dumps({"workload": "qwen-edit-2509 1024^2 + 1 cond image, one step", "ms_per_step": s * 1e3,
                      "steps_per_s": 1 / s, "model_tflops": 167.4 / s, "frac_2.5PF": 167.4 / s / 2500,
                      "kernels": kernels(fn)}), flush=True)
    del m
    torch.cuda.empty_cache()


def wan():
    from apex_studio_amd.wan import WanTransformer3DModel
    m = WanTransformer3DModel(device=DEV, dtype=torch.bfloat16).init_synthetic(2)
    m.pack()
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(1, 16, 21, 90, 160, generator=g, device=DEV)
    enc = torch.randn(1, 512, 4096, generator=g, device=DEV).to(torch.bfloat16)
    t = torch.tensor([500.0], device=DEV)
    fn = lambda: m(hidden_states=x, timestep=t, encoder_hidden_states=enc, return_dict=False)[0]  # noqa: E731
    s = timed(fn, n=1, warm=1)
    print(json.dumps({"workload": "wan2.2-a14b 720p x 81f, one expert forward", "s_per_step": s,
                      "model_tflops": 6520.0 / s, "frac_2.5PF": 6520.0 / s / 2500, "kernels": kernels(fn)}),
          flush=True)
    del m
    torch.cuda.empty_cache()


def vae():
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    v = AutoencoderKLWan(device=DEV, dtype=torch.bfloat16)
    g = torch.Generator(device=DEV).manual_seed(3)
    for n, p in v.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(1.0)
        elif n.endswith("bias"):
            p.data.zero_()
        else:
            p.data.copy_((torch.randn(p.shape, generator=g, device=DEV) / p[0].numel() ** 0.5).to(p.dtype))
    v.enable_tiling()
    z = torch.randn(1, 16, 21, 90, 160, generator=g, device=DEV).to(torch.bfloat16)
    fn = lambda: v.decode(z, return_dict=False)[0]  # noqa: E731
    s = timed(fn, n=1, warm=1)
    print(json.dumps({"workload": "wan 3D-VAE decode 21x90x160 -> 81x720x1280, 4x7 tiles", "s_per_decode": s,
                      "algorithmic_tflops_untiled": 632.0 / s, "kernels": kernels(fn)}), flush=True)


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["qwen", "vae", "wan"]):
        {"qwen": qwen, "wan": wan, "vae": vae}[w]()
