#!/usr/bin/env python
"""A/B: two Flux-1024 clips on ONE GPU, back to back on one stream vs concurrently on two HIP streams (one resident
model, per-stream workspaces).  The question is whether the tile-quantisation gaps of the B=1 step (216/648/864 GEMM
tiles and 432 attention workgroups on 256 CUs) can be filled by a second clip, given that the chip is power-bound.
Interleaved rounds in one process; prints one JSON line.  Not the headline metric (BASELINE's config is B=1, one
clip per GPU): a render-queue option.  `MODE=batch`: one B = 2 Flux forward with `model.batch_streams` 1 vs 2; `WORKLOAD=qwen`: the
same for Qwen-Image; `WORKLOAD=qwen_cfg`: Qwen-Image true-CFG steps with `engine.cfg_streams` off / on."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_studio_amd  # noqa: E402,F401
from apex_studio_amd.engine_flux import calculate_shift, latent_image_ids  # noqa: E402
from apex_studio_amd.flux import FluxTransformer2DModel  # noqa: E402
from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FLUX_DEV, S_IMG, S_TXT  # noqa: E402

DEV = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "8"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))


def qwen_batch():
    """B = 2 forward of the QwenImage-Edit step (S_img 8192 + S_txt 256 per image): model.batch_streams 1 vs 2."""
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    model = QwenImageTransformer2DModel(device=DEV, dtype=torch.bfloat16).init_synthetic(seed=4321)
    model.pack()
    g = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(1, 64, 64), (1, 64, 64)]
    kw = dict(hidden_states=torch.randn(2, 8192, 64, generator=g, device=DEV).to(torch.bfloat16),
              encoder_hidden_states=torch.randn(2, 256, 3584, generator=g, device=DEV).to(torch.bfloat16),
              encoder_hidden_states_mask=None, timestep=torch.tensor([0.7, 0.7], device=DEV, dtype=torch.bfloat16),
              img_shapes=[shapes, shapes], txt_seq_lens=[256, 256], return_dict=False)
    res, outs = {1: [], 2: []}, {}
    for _ in range(ROUNDS):
        for ns in (1, 2):
            model.batch_streams = ns
            outs[ns] = model(**kw)[0].clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                model(**kw)
            torch.cuda.synchronize()
            res[ns].append(round(1e3 * (time.perf_counter() - t0) / (2 * STEPS), 3))
    print(json.dumps({"workload": "qwenimage-edit-2509 forward, B = 2", "forwards": STEPS,
                      "ms_per_image_forward": {f"batch_streams={k}": v for k, v in res.items()},
                      "gain": round(min(res[1]) / min(res[2]) - 1.0, 4),
                      "results_identical": bool(torch.equal(outs[1], outs[2]))}))


def qwen_cfg():
    """QwenImage-Edit true-CFG steps (conditional + unconditional forward per step): engine.cfg_streams False vs True."""
    from apex_studio_amd.engine_qwenimage import QwenImageEditPlusEngine
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    model = QwenImageTransformer2DModel(device=DEV, dtype=torch.bfloat16).init_synthetic(seed=4321)
    eng = QwenImageEditPlusEngine(model)
    g = torch.Generator(device=DEV).manual_seed(5)
    lat = torch.randn(1, 4096, 64, generator=g, device=DEV).to(torch.bfloat16)
    cond = torch.randn(1, 4096, 64, generator=g, device=DEV).to(torch.bfloat16)
    pe = torch.randn(1, 256, 3584, generator=g, device=DEV).to(torch.bfloat16)
    ne = torch.randn(1, 200, 3584, generator=g, device=DEV).to(torch.bfloat16)
    shapes = [[(1, 64, 64), (1, 64, 64)]]
    res, outs = {False: [], True: []}, {}
    for _ in range(ROUNDS + 1):
        for st in (False, True):
            eng.cfg_streams = st
            ts = eng.scheduler.set_timesteps(sigmas=torch.linspace(1.0, 1.0 / STEPS, STEPS).tolist(), mu=0.8, device=DEV)
            eng.scheduler.set_begin_index(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs[st] = eng.base_denoise(lat, ts, pe, shapes, image_latents=cond, negative_prompt_embeds=ne, true_cfg_scale=4.0,
                                        use_cfg_guidance=True)
            torch.cuda.synchronize()
            res[st].append(round(1e3 * (time.perf_counter() - t0) / STEPS, 3))
    print(json.dumps({"workload": "qwenimage-edit-2509 true-CFG step (cond + uncond forward, S_img 8192, text 256 / 200)",
                      "steps": STEPS, "ms_per_step": {f"cfg_streams={k}": v[1:] for k, v in res.items()},
                      "gain": round(min(res[False][1:]) / min(res[True][1:]) - 1.0, 4),
                      "results_identical": bool(torch.equal(outs[False], outs[True]))}))


def main():
    if os.environ.get("WORKLOAD") == "qwen":
        return qwen_batch()
    if os.environ.get("WORKLOAD") == "qwen_cfg":
        return qwen_cfg()
    model = FluxTransformer2DModel(**FLUX_DEV, device=DEV, dtype=torch.bfloat16).init_synthetic(seed=1234)
    model.pack()
    img_ids = latent_image_ids(64, 64).to(DEV)
    txt_ids = torch.zeros(S_TXT, 3, device=DEV)
    guidance = torch.full([1], 3.5, device=DEV, dtype=torch.float32)

    class Clip:
        def __init__(self, seed):
            g = torch.Generator(device=DEV).manual_seed(seed)
            self.lat0 = torch.randn(1, S_IMG, 64, generator=g, device=DEV).to(torch.bfloat16)
            self.enc = torch.randn(1, S_TXT, 4096, generator=g, device=DEV).to(torch.bfloat16)
            self.pooled = torch.randn(1, 768, generator=g, device=DEV).to(torch.bfloat16)
            self.sched = FlowMatchEulerDiscreteScheduler.flux_dev()
            self.stream = torch.cuda.Stream(device=DEV)
            self.reset()

        def reset(self):
            n = STEPS + 2
            self.ts = self.sched.set_timesteps(sigmas=torch.linspace(1.0, 1.0 / n, n).tolist(),
                                               mu=calculate_shift(S_IMG), device=DEV)
            self.sched.set_begin_index(0)
            self.lat = self.lat0.clone()
            self.i = 0

        def step(self):
            t = self.ts[self.i]
            v = model(hidden_states=self.lat, timestep=t.expand(1).to(self.lat.dtype) / 1000, guidance=guidance,
                      pooled_projections=self.pooled, encoder_hidden_states=self.enc, txt_ids=txt_ids,
                      img_ids=img_ids, return_dict=False)[0]
            self.lat = self.sched.step(v, t, self.lat, return_dict=False)[0]
            self.i += 1

    if os.environ.get("MODE") == "batch":      # one B=2 forward (num_images = 2): model.batch_streams 1 vs 2
        g = torch.Generator(device=DEV).manual_seed(5)
        kw = dict(hidden_states=torch.randn(2, S_IMG, 64, generator=g, device=DEV).to(torch.bfloat16),
                  encoder_hidden_states=torch.randn(2, S_TXT, 4096, generator=g, device=DEV).to(torch.bfloat16),
                  pooled_projections=torch.randn(2, 768, generator=g, device=DEV).to(torch.bfloat16),
                  timestep=torch.tensor([0.7, 0.7], device=DEV, dtype=torch.bfloat16), guidance=guidance.expand(2),
                  txt_ids=txt_ids, img_ids=img_ids, return_dict=False)
        res, outs = {1: [], 2: []}, {}
        for _ in range(ROUNDS):
            for ns in (1, 2):
                model.batch_streams = ns
                outs[ns] = model(**kw)[0].clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(STEPS):
                    model(**kw)
                torch.cuda.synchronize()
                res[ns].append(round(1e3 * (time.perf_counter() - t0) / (2 * STEPS), 3))
        print(json.dumps({"workload": "flux-dev 1024x1024 forward, B = 2 (num_images 2)", "forwards": STEPS,
                          "ms_per_image_forward": {f"batch_streams={k}": v for k, v in res.items()},
                          "gain": round(min(res[1]) / min(res[2]) - 1.0, 4),
                          "results_identical": bool(torch.equal(outs[1], outs[2]))}))
        return

    a, b = Clip(1), Clip(2)

    def run(concurrent):
        a.reset()
        b.reset()
        sa = a.stream
        sb = b.stream if concurrent else a.stream
        for c, s in ((a, sa), (b, sb)):          # one warm-up step each
            with torch.cuda.stream(s):
                c.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if concurrent:
            for _ in range(STEPS):
                with torch.cuda.stream(sa):
                    a.step()
                with torch.cuda.stream(sb):
                    b.step()
        else:
            for c in (a, b):
                with torch.cuda.stream(sa):
                    for _ in range(STEPS):
                        c.step()
        t_cpu = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, t_cpu, a.lat.float().clone(), b.lat.float().clone()

    res = {"seq": [], "conc": [], "cpu_enqueue_ms_per_step": []}
    ref = None
    same = True
    for _ in range(ROUNDS):
        for mode in ("seq", "conc"):
            dt, t_cpu, la, lb = run(mode == "conc")
            res[mode].append(round(1e3 * dt / (2 * STEPS), 3))
            if mode == "conc":
                res["cpu_enqueue_ms_per_step"].append(round(1e3 * t_cpu / (2 * STEPS), 3))
            if ref is None:
                ref = (la, lb)
            else:
                same = same and torch.equal(ref[0], la) and torch.equal(ref[1], lb)
    best = {m: min(res[m]) for m in ("seq", "conc")}
    print(json.dumps({"workload": "flux-dev 1024x1024, two clips on one GPU", "steps_per_clip": STEPS,
                      "ms_per_step_per_clip": res, "best": best,
                      "aggregate_steps_per_sec": {m: round(1e3 / best[m], 3) for m in best},
                      "gain": round(best["seq"] / best["conc"] - 1.0, 4),
                      "results_identical_across_modes": bool(same)}))


if __name__ == "__main__":
    main()
