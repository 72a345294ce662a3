#!/usr/bin/env python
"""Headline benchmark: Flux-Dev 1024x1024 denoise steps/s on MI355X (BASELINE.json configs[1]).

A "step" = one FluxTransformer2DModel forward (19 double + 38 single MM-DiT blocks, S_img 4096 +
S_txt 512, 24x128 heads, bf16, B=1, no CFG) + one FlowMatch-Euler scheduler.step, on synthetic
latents / prompt embeddings and random-init weights of the FLUX.1-dev architecture (no network for
checkpoints).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, each denoising its OWN clip (the reference's only multi-GPU mechanism:
one engine-runner actor per GPU, apps/api/src/api/ray_tasks.py:181-306) -> weak scaling, no collective
inside a step; RCCL is used once, before the timed region, to broadcast the shared prompt embeddings
and a stand-in for the shared text-encoder/VAE weights (render_queue.broadcast_shared).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     achieved TFLOP/s of the dominant kernel (gemm_bf16_kernel) = algorithmic 2MNK flops of
               its launches / their summed HIP-event durations, measured live over extra profiled steps
  cpu_baseline the CPU oracle (fp32 PyTorch restatement of the reference path) timed on this box's
               host cores on a bounded sample (1 double + 1 single block at full width/sequence),
               extrapolated to 19 + 38 blocks
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLUX_DEV = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                guidance_embeds=True, axes_dims_rope=(16, 56, 56))
S_IMG, S_TXT = 4096, 512
# algorithmic FLOPs of one step: 2MNK per GEMM + 4 H Sq Sk D per attention
STEP_TFLOP = 74.36
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=str, default="", help="debug: 'D,S' block counts (invalid as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--broadcast-mib", type=int, default=1024)
    return ap.parse_args()


def cpu_baseline():
    """Oracle on host cores: one double + one single block at full width and sequence, fp32."""
    from oracle import flux as OF
    from oracle import layers as OL
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    dim, H = 3072, 24
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, S_IMG, dim, generator=g)
    ctx = torch.randn(1, S_TXT, dim, generator=g)
    temb = torch.randn(1, dim, generator=g)
    ids = torch.cat((torch.zeros(S_TXT, 3), OF.latent_image_ids(64, 64)), dim=0)
    rope = OF.flux_pos_embed(ids, (16, 56, 56))
    dbl = OF.FluxTransformerBlock(dim, H, 128).eval()
    sgl = OF.FluxSingleTransformerBlock(dim, H, 128).eval()
    with torch.no_grad():
        t0 = time.perf_counter()
        dbl(x, ctx, temb, rope, OL.FP32)
        t1 = time.perf_counter()
        sgl(x, ctx, temb, rope, OL.FP32)
        t2 = time.perf_counter()
    t_step = 19 * (t1 - t0) + 38 * (t2 - t1)
    return {
        "value": 1.0 / t_step, "unit": "steps/s", "cores": ncores, "kind": "port",
        "sample": (f"oracle fp32 (PyTorch CPU restatement of the reference path), 1 double block "
                   f"({t1 - t0:.2f} s) + 1 single block ({t2 - t1:.2f} s) at full width 3072 / S=4608, "
                   f"extrapolated x19 / x38; embedders and final layer (<0.1% of FLOPs) excluded"),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lib
    from apex_studio_amd.flux import FluxTransformer2DModel
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from apex_studio_amd import render_queue
    from apex_studio_amd.engine_flux import latent_image_ids, calculate_shift

    cfg = dict(FLUX_DEV)
    if args.layers:
        d, s = (int(v) for v in args.layers.split(","))
        cfg.update(num_layers=d, num_single_layers=s)
    model = FluxTransformer2DModel(**cfg, device=dev, dtype=torch.bfloat16).init_synthetic(seed=1234 + rank)
    model.pack()

    # this rank's clip: its own noise; the prompt embeddings are shared -> broadcast from rank 0
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    latents = torch.randn(1, S_IMG, 64, generator=g, device=dev).to(torch.bfloat16)
    gs = torch.Generator(device=dev).manual_seed(7)
    enc = torch.randn(1, S_TXT, 4096, generator=gs, device=dev).to(torch.bfloat16)
    pooled = torch.randn(1, 768, generator=gs, device=dev).to(torch.bfloat16)
    bcast = None
    if distributed:
        shared = torch.empty(args.broadcast_mib << 20, dtype=torch.uint8, device=dev)
        bcast = render_queue.broadcast_shared([enc, pooled, shared], src=0)
        del shared
    img_ids = latent_image_ids(64, 64).to(dev)
    txt_ids = torch.zeros(S_TXT, 3, device=dev)
    guidance = torch.full([1], 3.5, device=dev, dtype=torch.float32)

    total = args.warmup + args.steps
    sched = FlowMatchEulerDiscreteScheduler.flux_dev()
    sig = torch.linspace(1.0, 1.0 / total, total).tolist()
    timesteps = sched.set_timesteps(sigmas=sig, mu=calculate_shift(S_IMG), device=dev)
    sched.set_begin_index(0)

    def step(i, lat):
        t = timesteps[i]
        ts = t.expand(1).to(lat.dtype)
        v = model(hidden_states=lat, timestep=ts / 1000, guidance=guidance, pooled_projections=pooled,
                  encoder_hidden_states=enc, txt_ids=txt_ids, img_ids=img_ids, return_dict=False)[0]
        return sched.step(v, t, lat, return_dict=False)[0]

    for i in range(args.warmup):
        latents = step(i, latents)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        latents = step(i, latents)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(latents.float()).all().item())

    roofline = None
    kernels = {}
    if rank == 0 and not args.no_roofline:
        nprof = min(3, args.steps)
        sched.set_begin_index(0)
        sched._step_index = None
        lib.prof_reset()
        lib.prof_enable(True)
        lat = latents
        for i in range(nprof):
            lat = step(i, lat)
        prof = lib.prof_read()
        lib.prof_enable(False)
        lib.prof_reset()
        for name, r in prof.items():
            if r["launches"]:
                kernels[name] = {"ms_per_step": r["ms"] / nprof, "launches_per_step": r["launches"] / nprof,
                                 "avg_launch_us": 1e3 * r["ms"] / r["launches"],
                                 "tflops": (r["flops"] / (r["ms"] * 1e-3) / 1e12) if r["flops"] else None,
                                 "gbps": (r["bytes"] / (r["ms"] * 1e-3) / 1e9) if r["bytes"] else None}
        gk = prof["gemm"]
        ach = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                    "avg_launch_us": 1e3 * gk["ms"] / gk["launches"],
                    "launches_per_step": gk["launches"] / nprof,
                    "algorithmic_tflop_per_step": gk["flops"] / nprof / 1e12}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        full = not args.layers
        out = {
            "metric": "denoise_steps_per_sec", "value": args.gpus * args.steps / elapsed, "unit": "steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "flux-dev-1024x1024 denoise step (19 double + 38 single MM-DiT blocks, "
                                   "S_img 4096 + S_txt 512, B=1, no CFG) + FlowMatch-Euler step"
                       if full else f"DEBUG reduced depth {args.layers} (not a valid result)",
                       "clips_in_flight": args.gpus, "parallelism": f"clip-per-gpu x{args.gpus}",
                       "step_tflop": STEP_TFLOP if full else None},
            "model_tflops_per_gpu": (STEP_TFLOP / (ms_per_step * 1e-3)) if full else None,
            "mfma_utilisation_step": (STEP_TFLOP / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS) if full else None,
            "finite": finite,
            "roofline": roofline, "kernels": kernels, "broadcast": bcast,
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
