#!/usr/bin/env python
"""In-process interleaved A/B of the Flux-Dev 1024^2 denoise step (one resident model, same box, same run): each ARM is a set of
switches applied before a timed run of N steps, arms alternate for ROUNDS rounds.

  ARMS="base;modtable=0;ln.wave=2"   (';' separates arms, ',' separates switches inside an arm)
     modtable=0|1      per-clip modulation table (begin_schedule) off / on            [default on]
     <tune key>=<int>  apexmi_tune_set                                                 (reset to DEFAULTS after the arm)
     env:NAME=VALUE    os.environ for the arm (variables read per call only)
     attr:NAME=INT     model attribute for the arm (e.g. attr:fuse_qkv=0)
  CLK=1                also report the live shader clock over the GEMM K-loops of each run (apexmi_clk_*)
  STEPS=12 ROUNDS=3

Prints one JSON line per (round, arm) and a summary (median ms/step per arm, bit-identity of the final latents across arms)."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd import lib  # noqa: E402
from apex_studio_amd.engine_flux import calculate_shift, latent_image_ids  # noqa: E402
from apex_studio_amd.flux import FluxTransformer2DModel  # noqa: E402
from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402

DEV = "cuda"
DEFAULTS = {"ln.wave": 1, "gemm.group_m": 6, "gemm.large": 7, "gemm.tail": 2, "gemm.tail_max": 96, "gemm.x288": 0, "gemm.x384": 1, "gemm.x384_qkv": 1, "gemm.small_max": 112}      # shipped values of the keys an arm may set (restored after the arm)
ARMS = [a for a in os.environ.get("ARMS", "base;modtable=0").split(";") if a]
STEPS = int(os.environ.get("STEPS", "12"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
CLK = os.environ.get("CLK", "0") == "1"
FLUX_DEV = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128, num_attention_heads=24,
                joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))


def main():
    model = FluxTransformer2DModel(**FLUX_DEV, device=DEV, dtype=torch.bfloat16).init_synthetic(seed=1234)
    model.pack()
    g = torch.Generator(device=DEV).manual_seed(100)
    lat0 = torch.randn(1, 4096, 64, generator=g, device=DEV).to(torch.bfloat16)
    enc = torch.randn(1, 512, 4096, generator=g, device=DEV).to(torch.bfloat16)
    pooled = torch.randn(1, 768, generator=g, device=DEV).to(torch.bfloat16)
    img_ids, txt_ids = latent_image_ids(64, 64).to(DEV), torch.zeros(512, 3, device=DEV)
    guidance = torch.full([1], 3.5, device=DEV, dtype=torch.float32)
    sched = FlowMatchEulerDiscreteScheduler.flux_dev()

    def run(table):
        ts = sched.set_timesteps(sigmas=torch.linspace(1.0, 1.0 / STEPS, STEPS).tolist(), mu=calculate_shift(4096), device=DEV)
        sched.set_begin_index(0)
        lat = lat0
        torch.cuda.synchronize()
        if CLK:
            lib.clk_enable(True)
        t0 = time.perf_counter()
        if table:
            model.begin_schedule(torch.stack([t.expand(1).to(lat.dtype) / 1000 for t in ts]), guidance, pooled)
        for i, t in enumerate(ts):
            v = model(hidden_states=lat, timestep=t.expand(1).to(lat.dtype) / 1000, guidance=guidance, pooled_projections=pooled,
                      encoder_hidden_states=enc, txt_ids=txt_ids, img_ids=img_ids,
                      joint_attention_kwargs={"modulation_step": i} if table else None, return_dict=False)[0]
            lat = sched.step(v, t, lat, return_dict=False)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        model.end_schedule()
        ghz = None
        if CLK:
            ghz = lib.clk_read()["ghz"]
            lib.clk_enable(False)
        return 1e3 * dt / STEPS, lat, ghz

    run(True)
    res = {a: [] for a in ARMS}
    finals, clk = {}, {}
    for r in range(ROUNDS):
        for arm in ARMS:
            table, envs, keys, attrs = True, {}, [], {}
            for sw in ([] if arm == "base" else arm.split(",")):
                k, v = sw.split("=", 1)
                if k == "modtable":
                    table = v != "0"
                elif k.startswith("attr:"):
                    attrs[k[5:]] = getattr(model, k[5:])
                    setattr(model, k[5:], type(attrs[k[5:]])(int(v)))
                elif k.startswith("env:"):
                    envs[k[4:]] = os.environ.get(k[4:])
                    os.environ[k[4:]] = v
                else:
                    assert k in DEFAULTS, f"add the shipped value of {k} to DEFAULTS"
                    lib.tune_set(k, int(v))
                    keys.append(k)
            ms, lat, ghz = run(table)
            for k in keys:
                lib.tune_set(k, DEFAULTS[k])
            for k, v in attrs.items():
                setattr(model, k, v)
            for k, v in envs.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            res[arm].append(ms)
            finals.setdefault(arm, lat.clone())
            clk.setdefault(arm, []).append(ghz)
            print(json.dumps({"round": r, "arm": arm, "ms_per_step": ms, "gemm_clock_ghz": ghz}), flush=True)
    base = finals[ARMS[0]]
    print(json.dumps({"steps": STEPS, "rounds": ROUNDS,
                      "median_ms": {a: statistics.median(v) for a, v in res.items()},
                      "min_ms": {a: min(v) for a, v in res.items()},
                      "gemm_clock_ghz": {a: (statistics.median(v) if CLK else None) for a, v in clk.items()},
                      "final_latents_equal_to_first_arm": {a: bool(torch.equal(base, f)) for a, f in finals.items()}}), flush=True)


if __name__ == "__main__":
    main()
