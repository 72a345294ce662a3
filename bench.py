#!/usr/bin/env python
"""Headline benchmark: denoise steps/s (+ s/clip) on MI355X — BASELINE.json metric.

Default workload = BASELINE.json configs[1]: Flux-Dev 1024x1024.  A "step" = one
FluxTransformer2DModel forward (19 double + 38 single MM-DiT blocks, S_img 4096 + S_txt 512, 24x128
heads, bf16, B=1, no CFG) + one FlowMatch-Euler scheduler.step on synthetic latents / prompt embeddings
and random-init weights of the FLUX.1-dev architecture (no network for checkpoints).  Inputs are resident
in HBM before the timed region.  After the timed steps one whole clip (28 steps + 2-D VAE decode to
1024x1024, bf16) is timed as `sec_per_clip`.  The K timed steps are ONE clip: its per-clip set-up (the modulation table of its K
steps, after the hand-over from the warm-up clip) runs inside the timed region; the warm-up clip is scheduled equally long and has
itself been handed over once, so that set-up pays no first-use cost.  APEX_BENCH_STEP_TIMES=1 prints the set-up's and every timed
step's GPU time to stderr (diagnostics: two events per step inside the timed region).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Other single-GPU BASELINE configs (not the default bench line):
  --workload flux512 Flux-Dev 512^2 (config 1's geometry, S 1024 + 512, full depth) on the GPU: the small-shape line that
                     prices the host launch path (`host_enqueue_ms_per_step`)
  --workload qwen    QwenImage-Edit-2509 1024^2 + one 1024^2 condition image (config 3), steps/s
  --workload wan     Wan-2.2 A14B 720p x 81 frames, one expert forward + UniPC step (config 4), steps/s;
                     with --clip also 30 steps with the expert switch + tiled 3-D VAE decode (minutes)
  --workload hunyuan HunyuanVideo-1.5 480p x 121 frames, one forward + FlowMatch-Euler step (not a BASELINE config)
  --workload queue   config 5: 4 Flux-1024^2 clips + 4 Wan-720p clips sharded one clip per GPU
                     (--queue-wan-steps shortens the Wan clips; clips/hour, makespan)

N > 1: one process per GPU, each denoising its OWN clip (the reference's only multi-GPU mechanism: one
engine-runner actor per GPU, apps/api/src/api/ray_tasks.py:181-306) -> weak scaling, no collective inside
a step; RCCL is used once, before the timed region, to broadcast the weights every clip shares — the text encoders
and the VAE of the workload (Flux: T5-XXL + CLIP-L + 2-D VAE; Wan: UMT5-XXL + 3-D VAE), initialised on rank 0 only —
and the shared prompt embeddings (render_queue.broadcast_parameters / broadcast_shared); every rank then encodes the
same token ids and the outputs are compared bit for bit ("verified" in the JSON).
Started WITHOUT torchrun, `bench.py --gpus N` (N > 1) re-executes itself under torch.distributed.run with N ranks and
refuses to run when fewer than N GPUs are visible: an N-GPU number only ever comes from N processes.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     achieved TFLOP/s of the dominant kernel (gemm_bf16_kernel) = algorithmic 2MNK flops of its
               launches / their summed HIP-event durations, measured live over extra profiled steps
  cpu_baseline the CPU oracle (fp32 PyTorch restatement of the reference path) timed on this box's host
               cores on a bounded sample (1 double + 1 single block at full width/sequence), extrapolated
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
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # as the package sets it on import (apex-studio_amd/__init__.py); before torch

import torch  # noqa: E402

FLUX_DEV = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
                guidance_embeds=True, axes_dims_rope=(16, 56, 56))
S_IMG, S_TXT = 4096, 512
# algorithmic FLOPs of one step: 2MNK per GEMM + 4 H Sq Sk D per attention
# flux512 (config 1's geometry on the GPU: S 1024 + 512): every GEMM is linear in S (59.506 x 1536 / 4608 = 19.835), attention
# 4 x 24 x 1536^2 x 128 x 57 = 1.652
WAN_VAE_ALGORITHMIC_TFLOP = 632.0      # the untiled 720p x 81-frame decode (the reference's tiled execution ~ x1.78)
WAN_VAE_ALGORITHMIC_BYTES = 0.38e12    # minimal bf16 activation traffic with norm / activation fused
STEP_TFLOP = {"flux": 74.36, "flux512": 21.49, "qwen": 167.4, "wan": 6520.0, "hunyuan": 1394.9}
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["flux", "flux512", "qwen", "wan", "hunyuan", "queue"], default="flux")
    ap.add_argument("--layers", type=str, default="", help="debug: 'D,S' block counts (invalid as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the whole-clip (28 steps + decode) timing")
    ap.add_argument("--no-mod-table", action="store_true",
                    help="A/B: per-step AdaLN GEMVs instead of the per-clip modulation table (flux, qwen)")
    ap.add_argument("--clip", action="store_true", help="wan: also time a whole 30-step clip + decode")
    ap.add_argument("--fp8", action="store_true",
                    help="wan: block Linear weights RESIDENT as float8_e4m3fn + per-tensor scale_weight (the layout the Wan-2.2 manifest's "
                         "default files ship, R/manifest/video/wan-2.2-a14b-text-to-video-1.0.0.v1.yml:108-116), dequantised per call")
    ap.add_argument("--queue-wan-steps", type=int, default=30)
    ap.add_argument("--no-shared-weights", action="store_true",
                    help="N>1: broadcast only the prompt embeddings, not the text-encoder/VAE weights")
    ap.add_argument("--exchange-timeout", type=float, default=600.0,
                    help="N>1: seconds the weight exchange step (after the timed region) may take before the line is printed without it")
    ap.add_argument("--no-wan", action="store_true", help="N=1 flux: skip the Wan-2.2 720p half of the headline metric")
    ap.add_argument("--tune", type=str, default="", help="debug A/B: comma list of key=value for apexmi_tune_set")
    return ap.parse_args()


def cpu_info():
    """What the host cores ARE (BASELINE.md §3 asks for model, socket count and core count next to the CPU number)."""
    model, phys, cores, logical = "unknown", set(), set(), 0
    try:
        pid = "0"
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
                phys.add(v)
            elif k == "core id":
                cores.add((pid, v))
    except OSError:
        pass
    return {"cpu_model": model, "sockets": max(len(phys), 1), "physical_cores": len(cores) or None,
            "logical_cpus": logical or (os.cpu_count() or 1)}


def best_threads(fn, budget_s=20.0):
    """The thread count the CPU baseline runs at: all logical CPUs is NOT the fastest on a 2-socket SMT host (oversubscribed
    oneDNN / OpenMP teams: the 256-thread run of r4 was 5-6 x slower than 32 threads on 2 x EPYC 9575F).  `fn()` = one bounded piece
    of the workload; tried at {1/4, 1/8, 1/2, all of the physical cores, logical CPUs} until the budget is spent; returns
    (threads, {threads: seconds})."""
    info = cpu_info()
    logical = os.cpu_count() or 1
    phys = info["physical_cores"] or logical
    cands = []
    for c in (max(phys // 4, 1), max(phys // 8, 1), max(phys // 2, 1), phys, logical):
        if 1 <= c <= logical and c not in cands:
            cands.append(c)
    seen, t_start = {}, time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        fn()
        seen[c] = time.perf_counter() - t0
        if time.perf_counter() - t_start > budget_s:
            break
    best = min(seen, key=seen.get)
    torch.set_num_threads(best)
    return best, seen


def cpu_baseline(s_img=None, side=64):
    """Oracle on host cores: one double + one single block at full width and sequence, fp32."""
    from oracle import flux as OF
    from oracle import layers as OL
    dim, H = 3072, 24
    S_IMG = s_img or globals()["S_IMG"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, S_IMG, dim, generator=g)
    ctx = torch.randn(1, S_TXT, dim, generator=g)
    temb = torch.randn(1, dim, generator=g)
    ids = torch.cat((torch.zeros(S_TXT, 3), OF.latent_image_ids(side, side)), dim=0)
    rope = OF.flux_pos_embed(ids, (16, 56, 56))
    dbl = OF.FluxTransformerBlock(dim, H, 128).eval()
    sgl = OF.FluxSingleTransformerBlock(dim, H, 128).eval()
    with torch.no_grad():
        ncores, sweep = best_threads(lambda: sgl(x, ctx, temb, rope, OL.FP32))
        t0 = time.perf_counter()
        dbl(x, ctx, temb, rope, OL.FP32)
        t1 = time.perf_counter()
        sgl(x, ctx, temb, rope, OL.FP32)
        t2 = time.perf_counter()
    t_step = 19 * (t1 - t0) + 38 * (t2 - t1)
    return {
        "value": 1.0 / t_step, "unit": "steps/s", "cores": ncores, "kind": "port", **cpu_info(),
        "sample": (f"oracle fp32 (PyTorch CPU restatement of the reference path), 1 double block "
                   f"({t1 - t0:.2f} s) + 1 single block ({t2 - t1:.2f} s) at full width 3072 / S={S_IMG + S_TXT}, "
                   f"extrapolated x19 / x38; embedders and final layer (<0.1% of FLOPs) excluded; torch threads = {ncores}, the "
                   f"fastest of the single-block sweep {({k: round(v, 2) for k, v in sweep.items()})} s"),
    }


def _cpu_rate(flops, seconds, step_tflop, cores, sample):
    rate = flops / seconds                                    # algorithmic FLOP/s the host cores sustain on the sample
    return {"value": rate / (step_tflop * 1e12), "unit": "steps/s", "cores": cores, "kind": "port", **cpu_info(), "sample": sample}


def cpu_baseline_qwen():
    """Oracle QwenImage block (fp32) at full width on a shortened sequence, scaled to the step by algorithmic FLOPs."""
    from oracle import qwenimage as OQ
    from oracle import layers as OL
    dim, H, s_img, s_txt = 3072, 24, 4096, 256
    g = torch.Generator().manual_seed(0)
    blk = OQ.QwenImageTransformerBlock(dim, H, 128).eval()
    img, txt, temb = torch.randn(1, s_img, dim, generator=g), torch.randn(1, s_txt, dim, generator=g), torch.randn(1, dim, generator=g)
    rope = OQ.qwen_rope_table(OQ.qwen_rope_positions([(1, 64, 64)], s_txt), (16, 56, 56))
    with torch.no_grad():
        ncores, sweep = best_threads(lambda: blk(img, txt, temb, rope, OL.FP32))
        dt = sweep[ncores]
    S = s_img + s_txt
    flops = 2.0 * S * 12 * dim * dim + 4.0 * S * S * dim
    return _cpu_rate(flops, dt, STEP_TFLOP["qwen"], ncores,
                     f"oracle fp32 QwenImage block at full width 3072, S = {s_img} + {s_txt} tokens ({dt:.2f} s, "
                     f"{flops / 1e12:.2f} TFLOP); steps/s = sustained FLOP/s / {STEP_TFLOP['qwen']} TFLOP per step; torch threads = {ncores}, the "
                     f"fastest of {({k: round(v, 2) for k, v in sweep.items()})} s")


def cpu_baseline_wan():
    """Oracle Wan block (fp32) at full width on a shortened clip, scaled to the step by algorithmic FLOPs."""
    from oracle import wan as OW
    from oracle import layers as OL
    dim, H, ffn, s_txt, grid = 5120, 40, 13824, 512, (3, 30, 52)
    S = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(0)
    blk = OW.WanTransformerBlock(dim, ffn, H).eval()
    x, ctx = torch.randn(1, S, dim, generator=g), torch.randn(1, s_txt, dim, generator=g)
    temb6 = torch.randn(1, 6, dim, generator=g)
    rope = OW.wan_rope_table(grid, 128)
    with torch.no_grad():
        ncores, sweep = best_threads(lambda: blk(x, ctx, temb6, rope, OL.FP32))
        dt = sweep[ncores]
    flops = 2.0 * S * (6 * dim * dim + 2 * dim * ffn) + 2.0 * s_txt * 2 * dim * dim + 4.0 * S * S * dim + 4.0 * S * s_txt * dim
    return _cpu_rate(flops, dt, STEP_TFLOP["wan"], ncores,
                     f"oracle fp32 Wan block at full width 5120 / ffn 13824 on a {grid} latent grid (S = {S}) + {s_txt} "
                     f"text tokens ({dt:.2f} s, {flops / 1e12:.2f} TFLOP); steps/s = sustained FLOP/s / "
                     f"{STEP_TFLOP['wan']} TFLOP per expert forward; torch threads = {ncores}, the fastest of "
                     f"{({k: round(v, 2) for k, v in sweep.items()})} s")


def kernel_source_sha256(src_file):
    """sha256 of a kernel translation unit: the .hip file followed by its generated / multi-include pieces in csrc/ (for
    attention.hip: attn_*.h, attn_*.inc = the generated loop of the w64 kernel), in sorted order.  tools/gpu_pmc*.sh record the same."""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "apex-studio_amd", "csrc")
    stem = {"attention.hip": "attn_"}.get(src_file)
    files = [os.path.join(csrc, src_file)]
    if stem:
        files += sorted(glob.glob(os.path.join(csrc, stem + "*.h")) + glob.glob(os.path.join(csrc, stem + "*.inc")))
    h = hashlib.sha256()
    for fn in files:
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(src_file, suffix):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc summary (tools/gpu_pmc*.sh; separate passes, cannot run inside
    this process): used ONLY if it was taken from the kernel source this binary was built from (sha256 recorded next to it);
    otherwise null — never a stale constant.  Returns (bytes per launch, source file, the record)."""
    import glob
    import hashlib
    if not suffix:
        return None, None, None
    src_hash = kernel_source_sha256(src_file)
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_{suffix}")), reverse=True):
        rec = json.load(open(pmc))
        if rec.get("source_sha256") == src_hash:
            return rec.get("traffic_bytes_per_launch"), "profiles/" + os.path.basename(pmc), rec
    return None, None, None


HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s peak (~6.3 TB/s achievable)


def decode_roofline(prof, seconds, pmc_suffix, algorithmic_tflop, algorithmic_bytes, what):
    """The VAE decode against both rooflines (north_star: "rocprof counters reporting achieved HBM GB/s and MFMA utilisation" for
    the VAE).  `prof` = the library's per-class records over ONE timed decode (class `gemm` carries the convolutions: 2 M Cout
    taps Cin flops and input + weight + output bytes per launch; `attention` the mid-block attention; the rest the norm /
    upsample / blend passes with their bytes).  executed = what the launches computed (tiled: overlapping tiles recompute their
    halos, R/src/vae/wan/model.py:1516-1623); algorithmic = the untiled decode.  `traffic` = HBM-side bytes of
    the whole decode from the hash-matched rocprofv3 --pmc record (tools/gpu_pmc_vae.sh), else null."""
    ex_flops = sum(v["flops"] for v in prof.values())
    ex_bytes = sum(v["bytes"] for v in prof.values())
    kernel_ms = sum(v["ms"] for v in prof.values())
    traffic, src, rec = None, None, None
    import glob
    import hashlib
    conv_hash = hashlib.sha256(open(os.path.join(ROOT, "apex-studio_amd", "csrc", "conv.hip"), "rb").read()).hexdigest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_{pmc_suffix}")), reverse=True):
        r = json.load(open(f))
        if r.get("source_sha256") == conv_hash:
            traffic, src, rec = r.get("traffic_bytes_per_decode"), "profiles/" + os.path.basename(f), r
            break
    ach = ex_flops / seconds / 1e12
    return {"what": what, "seconds": seconds, "launches": int(sum(v["launches"] for v in prof.values())),
            "executed_tflop": ex_flops / 1e12, "algorithmic_tflop": algorithmic_tflop,
            "executed_over_algorithmic": ex_flops / 1e12 / algorithmic_tflop if algorithmic_tflop else None,
            "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
            "frac_algorithmic": (algorithmic_tflop / seconds / PEAK_BF16_TFLOPS) if algorithmic_tflop else None,
            "kernel_ms_sum": kernel_ms,
            "hbm": {"launch_bytes": ex_bytes, "launch_gbps": ex_bytes / seconds / 1e9, "peak_gbps": HBM_PEAK_GBPS,
                    "frac_launch_bytes": ex_bytes / seconds / 1e9 / HBM_PEAK_GBPS,
                    "algorithmic_bytes": algorithmic_bytes, "traffic": traffic, "traffic_source": src,
                    "traffic_gbps": (traffic / seconds / 1e9) if traffic else None,
                    "traffic_over_algorithmic": (traffic / algorithmic_bytes) if traffic and algorithmic_bytes else None,
                    "traffic_over_launch_bytes": (traffic / ex_bytes) if traffic and ex_bytes else None,
                    "mfma_pipe_busy_conv_kernels": (rec or {}).get("mfma_pipe_busy_fraction_conv_kernels"),
                    "l2_hit_rate": (rec or {}).get("l2_hit_rate")},
            "note": "achieved = executed flops of every launch of ONE decode / its wall time (tiles run two at a time on side "
                    "streams, so the per-launch HIP-event durations in kernel_ms_sum overlap and exceed the wall time); launch_bytes "
                    "= input + weight + output bytes each launch must move if nothing stayed in cache (intermediates are written "
                    "and read back once per layer: the un-fused upper bound of useful traffic); traffic = measured HBM-side bytes "
                    "(2 x FETCH_SIZE + WRITE_SIZE summed over the decode's dispatches) from the hash-matched PMC record"}


def timed_decode(fn):
    """fn() twice (warm, then timed with the library's per-class records on): (output, seconds, records)."""
    from apex_studio_amd import lib
    out = None
    for rep in range(2):
        torch.cuda.synchronize()
        if rep == 1:
            lib.prof_reset()
            lib.prof_enable(True)
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    prof = lib.prof_read()
    lib.prof_enable(False)
    lib.prof_reset()
    return out, dt, prof


def wan_half(dev, cpu=True):
    """The other half of BASELINE.json's metric on the default line: Wan-2.2 A14B 720p x 81 frames (config 4), one
    expert, 1 warm-up + 2 timed [forward + UniPC step], then the tiled 3-D VAE decode (1 warm-up + 1 timed)."""
    from apex_studio_amd import lib
    from apex_studio_amd.schedulers import UniPCMultistepScheduler
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    from apex_studio_amd.wan import WanTransformer3DModel
    model = WanTransformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(seed=999)
    model.pack()
    g = torch.Generator(device=dev).manual_seed(300)
    lat = torch.randn(1, 16, 21, 90, 160, generator=g, device=dev)
    enc = torch.randn(1, 512, 4096, generator=torch.Generator(device=dev).manual_seed(9), device=dev).to(torch.bfloat16)
    sched = UniPCMultistepScheduler(shift=3.0)
    ts = sched.set_timesteps(30, device=dev)

    def step(i, x):
        v = model(hidden_states=x.to(torch.bfloat16), timestep=ts[i].expand(1), encoder_hidden_states=enc,
                  return_dict=False)[0]
        return sched.step(v.float(), ts[i], x, return_dict=False)[0]

    lat = step(0, lat)
    torch.cuda.synchronize()
    lib.prof_reset()
    lib.prof_enable(True)
    t0 = time.perf_counter()
    for i in (1, 2):
        lat = step(i, lat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    prof = lib.prof_read()
    lib.prof_enable(False)
    lib.prof_reset()
    finite = bool(torch.isfinite(lat).all().item())
    att = prof["attention"]
    del model
    torch.cuda.empty_cache()
    vae = synth_vae_init(AutoencoderKLWan(device=dev, dtype=torch.bfloat16), 6)
    vae.enable_tiling()
    z = vae.denormalize_latents(lat).to(torch.bfloat16)
    video, dec, dprof = timed_decode(lambda: vae.decode(z, return_dict=False)[0])
    # the library's records cost one HIP-event pair per launch; the bare time is taken once more without them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    video = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize()
    dec_bare = time.perf_counter() - t0
    decode = decode_roofline(dprof, dec, "pmc_conv.json", WAN_VAE_ALGORITHMIC_TFLOP, WAN_VAE_ALGORITHMIC_BYTES,
                             "Wan 3-D VAE, tiled 4 x 7 decode of [1,16,21,90,160] -> [1,3,81,720,1280]")
    decode["seconds_without_records"] = dec_bare
    dec = min(dec, dec_bare)
    tf = STEP_TFLOP["wan"]
    ach = att["flops"] / (att["ms"] * 1e-3) / 1e12 if att["ms"] else None
    traffic, traffic_src, rec = pmc_traffic("attention.hip", "pmc_attn_wan.json")
    alg = (rec or {}).get("algorithmic_bytes_per_launch")
    roof = {"bound": "mfma", "kernel": "attn_fwd_d128_w64_kernel", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_BF16_TFLOPS if ach else None, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (traffic / alg) if traffic and alg else None,
            "avg_launch_us": 1e3 * att["ms"] / att["launches"] if att["launches"] else None,
            "launches_per_step": att["launches"] / 2, "algorithmic_tflop_per_step": att["flops"] / 2 / 1e12,
            "note": "the dominant kernel of the Wan step (72 % of it): achieved = 4 H Sq Sk D flops of its launches / their summed "
                    "HIP-event durations over the 2 timed steps (events on the launch stream); traffic = HBM bytes per launch from "
                    "the hash-matched rocprofv3 --pmc pass, mean over 40 self-attention (S 75600) + 40 cross-attention (512 keys) "
                    "launches"}
    out = {"workload": "wan-2.2-a14b text-to-video 720p x 81 frames: one expert forward (40 blocks, S 75600 + 512 text "
                       "tokens, B=1, no CFG) + UniPC step; tiled 3-D VAE decode to [1,3,81,720,1280]",
           "steps_timed": 2, "ms_per_step": 1e3 * dt, "steps_per_sec": 1.0 / dt, "step_tflop": tf,
           "model_tflops": tf / dt, "mfma_utilisation_step": tf / dt / PEAK_BF16_TFLOPS,
           "attention": {"tflops": ach, "ms_per_step": att["ms"] / 2, "launches_per_step": att["launches"] / 2},
           "roofline": roof, "decode_s": dec, "decode": decode, "sec_per_clip_30_steps": 30 * dt + dec, "video": list(video.shape),
           "finite": finite and bool(torch.isfinite(video.float()).all().item())}
    del vae, video, z
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = cpu_baseline_wan()
    return out


def synth_vae_init(vae, seed):
    g = torch.Generator(device=vae.device).manual_seed(seed)
    for n, p in vae.named_parameters():
        if n.endswith("gamma") or (n.endswith("weight") and p.dim() == 1):
            p.data.fill_(1.0)
        elif n.endswith("bias"):
            p.data.zero_()
        else:
            p.data.copy_((torch.randn(p.shape, generator=g, device=p.device) / p[0].numel() ** 0.5).to(p.dtype))
    return vae


# ---- workloads: each returns (step_fn(i, state) -> state, state0, total_steps_setup_fn, extras) ----------------

def build_flux(args, dev, rank, total):
    from apex_studio_amd.flux import FluxTransformer2DModel
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from apex_studio_amd.engine_flux import latent_image_ids, calculate_shift
    cfg = dict(FLUX_DEV)
    if args.layers:
        d, s = (int(v) for v in args.layers.split(","))
        cfg.update(num_layers=d, num_single_layers=s)
    small = args.workload == "flux512"
    S_IMG, side, px, clip_steps = (1024, 32, 512, 4) if small else (4096, 64, 1024, 28)
    model = FluxTransformer2DModel(**cfg, device=dev, dtype=torch.bfloat16).init_synthetic(seed=1234 + rank)
    model.pack()
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    latents = torch.randn(1, S_IMG, 64, generator=g, device=dev).to(torch.bfloat16)
    gs = torch.Generator(device=dev).manual_seed(7)
    enc = torch.randn(1, S_TXT, 4096, generator=gs, device=dev).to(torch.bfloat16)
    pooled = torch.randn(1, 768, generator=gs, device=dev).to(torch.bfloat16)
    img_ids = latent_image_ids(side, side).to(dev)
    txt_ids = torch.zeros(S_TXT, 3, device=dev)
    guidance = torch.full([1], 3.5, device=dev, dtype=torch.float32)
    sched = FlowMatchEulerDiscreteScheduler.flux_dev()

    def reset(n):
        sig = torch.linspace(1.0, 1.0 / n, n).tolist()
        ts = sched.set_timesteps(sigmas=sig, mu=calculate_shift(S_IMG), device=dev)
        sched.set_begin_index(0)
        return ts

    ts_box = {"ts": reset(total)}

    sched_box = {"i0": None}

    def begin(i0, i1):
        """What `FluxT2IEngine.base_denoise` does when it enters its loop (engine_flux.py): steps [i0, i1) of the current
        timesteps are one clip whose AdaLN modulation vectors are computed in one pass over the projection weights.  The
        timed region calls this itself, so the table's cost sits INSIDE the measured time (charged to its first step)."""
        if sched_box.get("h") is not None:
            model.end_schedule(sched_box["h"])          # the previous clip is over (also: its rows matched its timesteps)
            sched_box["h"] = None
        if args.no_mod_table or i1 <= i0:
            sched_box["i0"] = None
            return
        ts = ts_box["ts"][i0:i1]
        sched_box["h"] = model.begin_schedule(torch.stack([t.expand(1).to(latents.dtype) / 1000 for t in ts]), guidance, pooled)
        sched_box["i0"] = i0

    def step(i, lat):
        t = ts_box["ts"][i]
        jkw = None if sched_box["i0"] is None else {"modulation_step": i - sched_box["i0"], "modulation_schedule": sched_box["h"]}
        v = model(hidden_states=lat, timestep=t.expand(1).to(lat.dtype) / 1000, guidance=guidance,
                  pooled_projections=pooled, encoder_hidden_states=enc, txt_ids=txt_ids, img_ids=img_ids,
                  joint_attention_kwargs=jkw, return_dict=False)[0]
        return sched.step(v, t, lat, return_dict=False)[0]
    step.begin = begin

    def clip():
        """One whole clip: 28 (flux512: 4) steps + unpack + 2-D VAE decode (engine/flux/t2i.py:251-255)."""
        from apex_studio_amd.engine_flux import unpack_latents
        from apex_studio_amd.vae_flux import AutoencoderKL
        vae = synth_vae_init(AutoencoderKL(device=dev, dtype=torch.bfloat16), 5)
        lat = torch.randn(1, S_IMG, 64, generator=g, device=dev).to(torch.bfloat16)
        for rep in range(2):        # first pass warms the VAE's packed weights
            ts_box["ts"] = reset(clip_steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = lat
            begin(0, clip_steps)
            for i in range(clip_steps):
                x = step(i, x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            z = vae.denormalize_latents(unpack_latents(x, px, px, 8))
            img = vae.decode(z, return_dict=False)[0]
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        # the decode once more under the library's per-class records: its roofline object (the 2-D twin of wan.decode)
        _, dsec, dprof = timed_decode(lambda: vae.decode(z, return_dict=False)[0])
        decode = decode_roofline(dprof, dsec, "pmc_conv_flux.json", None, None,
                                 f"Flux 2-D VAE decode of [1,16,{px // 8},{px // 8}] -> [1,3,{px},{px}] (untiled: executed = algorithmic)")
        decode["algorithmic_tflop"] = decode["executed_tflop"]
        decode["executed_over_algorithmic"] = 1.0
        decode["frac_algorithmic"] = decode["frac"]
        return {"sec_per_clip": t2 - t0, "denoise_s": t1 - t0, "decode_s": t2 - t1, "decode": decode, "steps": clip_steps,
                "image": list(img.shape), "finite": bool(torch.isfinite(img.float()).all().item())}

    label = (f"flux-dev-{px}x{px} denoise step (19 double + 38 single MM-DiT blocks, S_img {S_IMG} + S_txt 512, "
             "B=1, no CFG) + FlowMatch-Euler step") if not args.layers else \
        f"DEBUG reduced depth {args.layers} (not a valid result)"
    return step, latents, reset, [enc, pooled], clip, label


def build_qwen(args, dev, rank, total):
    from apex_studio_amd.qwenimage import QwenImageTransformer2DModel
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from apex_studio_amd.engine_flux import calculate_shift
    kw = {"num_layers": int(args.layers.split(",")[0])} if args.layers else {}     # debug depth (PMC passes), invalid as a result
    model = QwenImageTransformer2DModel(device=dev, dtype=torch.bfloat16, **kw).init_synthetic(seed=4321 + rank)
    model.pack()
    g = torch.Generator(device=dev).manual_seed(200 + rank)
    latents = torch.randn(1, 4096, 64, generator=g, device=dev).to(torch.bfloat16)
    cond = torch.randn(1, 4096, 64, generator=g, device=dev).to(torch.bfloat16)
    enc = torch.randn(1, 256, 3584, generator=torch.Generator(device=dev).manual_seed(8), device=dev).to(torch.bfloat16)
    shapes = [[(1, 64, 64), (1, 64, 64)]]
    sched = FlowMatchEulerDiscreteScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9,
                                            base_image_seq_len=256, max_image_seq_len=8192, shift_terminal=0.02)

    def reset(n):
        ts = sched.set_timesteps(sigmas=torch.linspace(1.0, 1.0 / n, n).tolist(),
                                 mu=calculate_shift(4096, 256, 8192, 0.5, 0.9), device=dev)
        sched.set_begin_index(0)
        return ts

    ts_box = {"ts": reset(total)}

    sched_box = {"i0": None}

    def begin(i0, i1):
        """`QwenImageEditPlusEngine.base_denoise` entering its loop: the modulation table of steps [i0, i1) (inside the timed region)."""
        if sched_box.get("h") is not None:
            model.end_schedule(sched_box["h"])
            sched_box["h"] = None
        if args.no_mod_table or i1 <= i0:
            sched_box["i0"] = None
            return
        sched_box["h"] = model.begin_schedule(torch.stack([t.expand(1).to(latents.dtype) / 1000 for t in ts_box["ts"][i0:i1]]))
        sched_box["i0"] = i0

    def step(i, lat):
        t = ts_box["ts"][i]
        x = torch.cat([lat, cond], dim=1)
        akw = None if sched_box["i0"] is None else {"modulation_step": i - sched_box["i0"], "modulation_schedule": sched_box["h"]}
        v = model(hidden_states=x, encoder_hidden_states=enc, encoder_hidden_states_mask=None,
                  timestep=t.expand(1).to(lat.dtype) / 1000, img_shapes=shapes, txt_seq_lens=[256],
                  attention_kwargs=akw, return_dict=False)[0][:, :4096]
        return sched.step(v, t, lat, return_dict=False)[0]
    step.begin = begin

    label = ("qwenimage-edit-2509 1024x1024 + one 1024x1024 condition image, denoise step (60 MM-DiT blocks, "
             "S_img 8192 + S_txt 256, B=1) + FlowMatch-Euler step")
    return step, latents, reset, [enc], None, label


@torch.no_grad()
def fp8_quantise_blocks(model):
    """Synthetic stand-in for a keep_fp8 load (weights.load_checkpoint_into(keep_fp8=True)): every block Linear weight becomes
    float8_e4m3fn + one `scale_weight` = amax / 448 (the fp8-scaled checkpoint layout, R/src/quantize/scaled_layer.py:390-552),
    then `_fp8_adopt()` fuses the projections' records and releases the bf16 storage."""
    from apex_studio_amd import ops
    for name, p in model.named_parameters():
        if model._fp8_resident_key(name) and p.dim() == 2:
            scale = (p.data.abs().amax().float() / 448.0).clamp_min(1e-12)
            q = (p.data.float() / scale).to(torch.float8_e4m3fn)
            p._fp8 = ops.Fp8Weight(q, scale.reshape(1))
    return model._fp8_adopt()


def build_wan(args, dev, rank, total):
    from apex_studio_amd.wan import WanTransformer3DModel
    from apex_studio_amd.schedulers import UniPCMultistepScheduler
    model = WanTransformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(seed=999 + rank)
    model.pack()
    if args.fp8:
        fp8_quantise_blocks(model)
    g = torch.Generator(device=dev).manual_seed(300 + rank)
    latents = torch.randn(1, 16, 21, 90, 160, generator=g, device=dev)
    enc = torch.randn(1, 512, 4096, generator=torch.Generator(device=dev).manual_seed(9), device=dev).to(torch.bfloat16)
    sched = UniPCMultistepScheduler(shift=3.0)

    def reset(n):
        return sched.set_timesteps(n, device=dev)

    ts_box = {"ts": reset(total)}

    def step(i, lat):
        t = ts_box["ts"][i]
        v = model(hidden_states=lat.to(torch.bfloat16), timestep=t.expand(1), encoder_hidden_states=enc,
                  return_dict=False)[0]
        return sched.step(v.float(), t, lat, return_dict=False)[0]

    def clip():
        from apex_studio_amd.engine_wan import WanT2VEngine
        from apex_studio_amd.vae_wan import AutoencoderKLWan
        vae = synth_vae_init(AutoencoderKLWan(device=dev, dtype=torch.bfloat16), 6)
        low = WanTransformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(seed=555 + rank)
        eng = WanT2VEngine(model, low, vae=vae)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        video = eng.run(prompt_embeds=enc, height=720, width=1280, duration=81, num_inference_steps=30,
                        generator=torch.Generator(device=dev).manual_seed(1))
        torch.cuda.synchronize()
        return {"sec_per_clip": time.perf_counter() - t0, "steps": 30, "video": list(video.shape),
                "finite": bool(torch.isfinite(video.float()).all().item())}

    label = ("wan-2.2-a14b text-to-video 720p x 81 frames, one expert forward (40 blocks, S 75600 + 512 text, B=1, "
             "no CFG) + UniPC step") + (f"; block weights resident fp8-scaled ({model._fp8_bytes / 2**30:.1f} GiB), dequantised per call"
                                        if args.fp8 else "")
    return step, latents, reset, [enc], (clip if args.clip else None), label


def build_hunyuan(args, dev, rank, total):
    """HunyuanVideo-1.5 480p x 121 frames T2V (not a BASELINE.json config): 54 MM-DiT blocks,
    d 2048 = 16 x 128, S_img 31*30*52 = 48360, condition tokens 1000 MLLM (300 valid) + 256 ByT5 (64 valid) + 729
    vision slots (masked for t2v).  273.7 TFLOP GEMM + 1121 TFLOP attention per forward."""
    from apex_studio_amd.hunyuan15 import HunyuanVideo15Transformer3DModel
    from apex_studio_amd.schedulers import FlowMatchEulerDiscreteScheduler
    model = HunyuanVideo15Transformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(seed=777 + rank)
    model.pack()
    g = torch.Generator(device=dev).manual_seed(400 + rank)
    latents = torch.randn(1, 32, 31, 30, 52, generator=g, device=dev)
    ge = torch.Generator(device=dev).manual_seed(9)
    enc = torch.randn(1, 1000, 3584, generator=ge, device=dev).to(torch.bfloat16)
    enc2 = torch.randn(1, 256, 1472, generator=ge, device=dev).to(torch.bfloat16)
    m1 = torch.zeros(1, 1000, device=dev)
    m1[:, :300] = 1
    m2 = torch.zeros(1, 256, device=dev)
    m2[:, :64] = 1
    img = torch.zeros(1, 729, 1152, device=dev, dtype=torch.bfloat16)
    zeros = torch.zeros(1, 33, 31, 30, 52, device=dev, dtype=torch.bfloat16)
    sched = FlowMatchEulerDiscreteScheduler(shift=7.0)

    def reset(n):
        return sched.set_timesteps(n, device=dev, sigmas=torch.linspace(1.0, 0.0, n + 1, dtype=torch.float64)[:-1])

    ts_box = {"ts": reset(total)}

    def step(i, lat):
        t = ts_box["ts"][i]
        x = torch.cat([lat.to(torch.bfloat16), zeros], dim=1)
        v = model(hidden_states=x, timestep=t.expand(1).to(torch.bfloat16), encoder_hidden_states=enc,
                  encoder_attention_mask=m1, encoder_hidden_states_2=enc2, encoder_attention_mask_2=m2, image_embeds=img,
                  return_dict=False)[0]
        return sched.step(v, t, lat, return_dict=False)[0]

    label = ("hunyuanvideo-1.5 text-to-video 480p x 121 frames, one forward (54 MM-DiT blocks, S 48360 latent + 1985 "
             "condition tokens, B=1, no CFG) + FlowMatch-Euler step")
    return step, latents, reset, [enc, enc2], None, label


def run_queue(args, dev, rank, world):
    """config 5: 4 Flux-1024^2 clips + 4 Wan-720p clips, one clip per GPU at a time (pulled by the free rank; LPT order as the seed)."""
    from apex_studio_amd import render_queue
    from apex_studio_amd.engine_flux import FluxT2IEngine
    from apex_studio_amd.engine_wan import WanT2VEngine
    from apex_studio_amd.flux import FluxTransformer2DModel
    from apex_studio_amd.wan import WanTransformer3DModel
    from apex_studio_amd.vae_flux import AutoencoderKL
    from apex_studio_amd.vae_wan import AutoencoderKLWan
    flux = FluxT2IEngine(FluxTransformer2DModel(**FLUX_DEV, device=dev, dtype=torch.bfloat16).init_synthetic(1),
                         decode_fn=None)
    fvae = synth_vae_init(AutoencoderKL(device=dev, dtype=torch.bfloat16), 5)
    flux.decode_fn = lambda z: fvae.decode(fvae.denormalize_latents(z), return_dict=False)[0]
    hi = WanTransformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(2)
    lo = WanTransformer3DModel(device=dev, dtype=torch.bfloat16).init_synthetic(3)
    wan = WanT2VEngine(hi, lo, vae=synth_vae_init(AutoencoderKLWan(device=dev, dtype=torch.bfloat16), 6))
    g = torch.Generator(device=dev).manual_seed(7)
    f_enc = torch.randn(1, S_TXT, 4096, generator=g, device=dev).to(torch.bfloat16)
    f_pool = torch.randn(1, 768, generator=g, device=dev).to(torch.bfloat16)
    w_enc = torch.randn(1, 512, 4096, generator=g, device=dev).to(torch.bfloat16)
    bcast = {"embeddings": render_queue.broadcast_shared([f_enc, f_pool, w_enc], src=0)}
    if world > 1 and not args.no_shared_weights:     # the queue's shared components: both families' encoders; the VAEs above
        from apex_studio_amd import text_encoders as TE
        bf = dict(device=dev, dtype=torch.bfloat16)
        enc_mods = [TE.T5EncoderModel({}, **bf), TE.CLIPTextModel({}, **bf), TE.UMT5EncoderModel({}, **bf)]
        if rank == 0:
            for i, m in enumerate(enc_mods):
                _synth_text_init(m, 40 + i)
        what = "T5-XXL + CLIP-L + UMT5-XXL text encoders + Flux 2-D VAE + Wan 3-D VAE"
        try:
            bcast["weights"] = dict(render_queue.broadcast_parameters(enc_mods + [fvae, wan.vae], src=0), what=what)
        except Exception as e:          # reported, never allowed to take the queue down (every rank built its own VAEs above)
            bcast["weights"] = {"what": what, "error": f"{type(e).__name__}: {e}"[:300]}
        del enc_mods
        torch.cuda.empty_cache()
    wan_cost = 6.0 * args.queue_wan_steps

    def runner(c):
        if c["kind"] == "flux":
            flux.run(prompt_embeds=f_enc, pooled_prompt_embeds=f_pool, height=1024, width=1024,
                     num_inference_steps=28, seed=c["seed"])
        else:
            wan.run(prompt_embeds=w_enc, height=720, width=1280, duration=81,
                    num_inference_steps=args.queue_wan_steps, seed=c["seed"])

    runner({"kind": "flux", "seed": 99})      # warm (packing, workspaces)
    # (1) the literal config 5: 8 clips, STRONG scaling — bounded by its longest clip once every GPU holds one
    clips = [{"kind": "flux", "seed": i, "cost": 2.5} for i in range(4)] + \
            [{"kind": "wan", "seed": 10 + i, "cost": wan_cost} for i in range(4)]
    res = render_queue.run_queue(clips, runner, dynamic=True)
    # (2) the same queue DEEPENED with the node (WEAK scaling: one Flux + one Wan clip per GPU) — the regime in which
    # clip-per-GPU sharding can show its N-fold throughput (north_star's >= 7.5x at 8 GPUs)
    weak_clips = [{"kind": "flux", "seed": 100 + i, "cost": 2.5} for i in range(world)] + \
                 [{"kind": "wan", "seed": 200 + i, "cost": wan_cost} for i in range(world)]
    weak = render_queue.run_queue(weak_clips, runner, dynamic=True)
    _flush_c_stdio()
    if rank == 0:
        print(json.dumps({
            "metric": "queue_clips_per_hour", "value": res["clips_per_hour"], "unit": "clips/h", "n_gpus": world,
            "steps": 1, "warmup": 1, "ms_per_step": 1e3 * res["makespan"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"8-clip render queue: 4x flux-dev 1024^2 (28 steps + decode) + 4x wan-2.2 "
                                   f"720p x 81f ({args.queue_wan_steps} steps, expert switch, tiled 3D-VAE decode), "
                                   f"one clip per GPU at a time", "parallelism": f"clip-per-gpu x{world}"},
            "world": world, "makespan_s": res["makespan"], "busy_s": res["busy"], "dispatch": res["dispatch"],
            "clip_rank": {str(k): v for k, v in sorted(res["clip_rank"].items())},
            "clip_seconds": {str(k): round(v, 3) for k, v in sorted(res["clip_seconds"].items())},
            "weak_scaling": {"clips": len(weak_clips), "what": "one flux-dev 1024^2 clip + one wan-2.2 720p clip PER GPU",
                             "clips_per_hour": weak["clips_per_hour"], "makespan_s": weak["makespan"],
                             "busy_s": weak["busy"]},
            "broadcast": bcast}), flush=True)


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, RCCL) and exit with their
    status.  Refuses when the node shows fewer than N GPUs — a single process never reports an N-GPU number."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("APEX_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node; an {args.gpus}-GPU "
                         f"result needs {args.gpus} ranks, one per GPU — refusing to run")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def _synth_text_init(m, seed):
    g = torch.Generator(device=m.device).manual_seed(seed)
    for n, p in m.named_parameters():
        if "norm" in n or n.endswith("ln_q.weight"):
            p.data.fill_(1.0)
        elif n.endswith("bias"):
            p.data.zero_()
        else:
            sc = (0.125 if n.endswith(".q.weight") else 1.0) / p.shape[-1] ** 0.5
            flat = p.data.view(-1)
            for i in range(0, flat.numel(), 1 << 26):
                k = min(1 << 26, flat.numel() - i)
                flat[i:i + k] = (torch.randn(k, generator=g, device=p.device) * sc).to(p.dtype)
    return m


def shared_weights(workload, dev, rank):
    """The components every clip of a queue shares (reference manifests: Flux yml:55,66,83; Wan yml:84,71): built on
    every rank, initialised (stand-in for loaded) on rank 0 only.  Returns (modules, label, probe) where probe(mods)
    encodes fixed token ids and returns a checksum tensor."""
    from apex_studio_amd import text_encoders as TE
    bf = dict(device=dev, dtype=torch.bfloat16)
    if workload in ("flux", "queue"):
        from apex_studio_amd.vae_flux import AutoencoderKL
        mods = [TE.T5EncoderModel({}, **bf), TE.CLIPTextModel({}, **bf), AutoencoderKL(**bf)]
        label = "T5-XXL + CLIP-L text encoders + Flux 2-D VAE"
    elif workload == "wan":
        from apex_studio_amd.vae_wan import AutoencoderKLWan
        mods = [TE.UMT5EncoderModel({}, **bf), AutoencoderKLWan(**bf)]
        label = "UMT5-XXL text encoder + Wan 3-D VAE"
    else:
        return None, None, None
    if rank == 0:
        for i, m in enumerate(mods[:-1]):
            _synth_text_init(m, 40 + i)
        synth_vae_init(mods[-1], 5)

    def probe(ms):
        out = []
        for m in ms[:-1]:
            S = 77 if isinstance(m, TE.CLIPTextModel) else 512
            ids = (torch.arange(S, device=dev) * 37 % 30000 + 3).reshape(1, S)
            out.append(m(input_ids=ids, attention_mask=torch.ones_like(ids)).last_hidden_state.double().sum())
        return torch.stack(out)
    return mods, label, probe


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: every GPU counted in the result must be "
                         f"one rank of this job")
    import torch.distributed as dist
    distributed = world > 1 or os.environ.get("APEX_FORCE_DIST") == "1"   # force: RCCL smoke test on one GPU
    # APEX_BENCH_SHARE_GPU=1 (DEBUG, invalid as a result): every rank on cuda:0 over gloo — rehearses the N > 1 control flow
    # (rank spawn, barriers, max-over-ranks, the exchange step's failure path: gloo has no CUDA scatter) on a one-GPU box
    share_gpu = os.environ.get("APEX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    for kv in filter(None, args.tune.split(",")):
        from apex_studio_amd import lib as _lib
        key, val = kv.split("=")
        _lib.tune_set(key, int(val))
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")        # no version banner on stdout next to the JSON line
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import apex_studio_amd  # noqa: F401
    from apex_studio_amd import lib, render_queue

    if args.workload == "queue":
        run_queue(args, dev, rank, world)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    total = args.warmup + args.steps
    build = {"flux": build_flux, "flux512": build_flux, "qwen": build_qwen, "wan": build_wan, "hunyuan": build_hunyuan}[args.workload]
    step, latents, reset, shared_inputs, clip_fn, label = build(args, dev, rank, total)

    bcast = None
    if distributed:   # shared prompt embeddings from rank 0 (small: plain broadcasts); the WEIGHT exchange follows the timed region
        bcast = {"embeddings": render_queue.broadcast_shared(list(shared_inputs), src=0),
                 "rccl_ranks": dist.get_world_size()}

    begin = getattr(step, "begin", lambda i0, i1: None)   # the engine's per-clip setup (Flux: the modulation table), see build_flux
    # the warm-up clip is SCHEDULED as long as the timed one (its per-clip table has the timed clip's size, so the allocator and the
    # kernels' first launches are warm — a resident engine renders clip after clip of one length) and runs its first W steps
    # — twice, so that the hand-over between two clips (end of the previous schedule, its consistency check) has run before the timed
    # region does it
    begin(0, min(total, max(args.warmup, args.steps)))
    begin(0, min(total, max(args.warmup, args.steps)))
    for i in range(args.warmup):
        latents = step(i, latents)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_marks = [] if os.environ.get("APEX_BENCH_STEP_TIMES") == "1" else None   # diagnostics: an event after begin() and after every step
    if step_marks is not None:
        step_marks.append(torch.cuda.Event(enable_timing=True)); step_marks[-1].record()
    begin(args.warmup, total)                             # INSIDE the timed region: the K timed steps are one clip
    if step_marks is not None:
        step_marks.append(torch.cuda.Event(enable_timing=True)); step_marks[-1].record()
    for i in range(args.warmup, total):
        latents = step(i, latents)
        if step_marks is not None:
            step_marks.append(torch.cuda.Event(enable_timing=True)); step_marks[-1].record()
    enqueued = time.perf_counter() - t0                   # host side done: every launch of the K steps is in the queue
    torch.cuda.synchronize()
    if step_marks is not None and rank == 0:
        print("[step times] begin %.3f ms; steps %s" % (step_marks[0].elapsed_time(step_marks[1]),
              " ".join("%.2f" % step_marks[k].elapsed_time(step_marks[k + 1]) for k in range(1, len(step_marks) - 1))), file=sys.stderr)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(latents.float()).all().item())
    # The Python + ctypes launch path, priced: OUTSIDE the timed region, from an idle queue, a burst of steps
    # short enough never to meet HIP's queue-depth back-pressure (~2000 launches in flight: the timed loop's own enqueue time
    # is mostly that wait once the host is a few steps ahead) is enqueued with no sync inside; its wall time is host work only.
    nburst = max(1, min(3, args.steps))
    reset(total)
    begin(0, nburst)
    lat_b = latents
    torch.cuda.synchronize()
    tb = time.perf_counter()
    for i in range(nburst):
        lat_b = step(i, lat_b)
    host_burst = (time.perf_counter() - tb) / nburst
    torch.cuda.synchronize()
    burst_total = (time.perf_counter() - tb) / nburst
    del lat_b

    def core_line(extra_bcast=None):
        """The driver's line from what is known right after the timed region (the watchdog below prints exactly this)."""
        ms = 1e3 * elapsed / args.steps
        full_ = not args.layers
        tf_ = STEP_TFLOP[args.workload]
        b = dict(bcast or {})
        if extra_bcast:
            b.update(extra_bcast)
        return {"metric": "denoise_steps_per_sec", "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": args.gpus,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": label, "clips_in_flight": args.gpus, "parallelism": f"clip-per-gpu x{args.gpus}",
                           "step_tflop": tf_ if full_ else None},
                "world": world, "rccl_ranks": world if distributed else 1,
                "model_tflops_per_gpu": (tf_ / (ms * 1e-3)) if full_ else None,
                "mfma_utilisation_step": (tf_ / (ms * 1e-3) / PEAK_BF16_TFLOPS) if full_ else None,
                "finite": finite, "broadcast": b or None, **({"debug_shared_gpu": True} if share_gpu else {}),
                "host_enqueue_ms_per_step": 1e3 * host_burst,
                "gpu_to_host_enqueue_ratio": (elapsed / args.steps) / host_burst if host_burst > 0 else None,
                "host_enqueue": {"burst_steps": nburst, "burst_ms_per_step_incl_gpu": 1e3 * burst_total,
                                 "timed_loop_enqueue_ms_per_step": 1e3 * enqueued / args.steps,
                                 "note": "host_enqueue_ms_per_step = wall time to enqueue one step's launches from an idle queue "
                                         "(no sync, burst outside the timed region); timed_loop_enqueue includes the HIP queue's "
                                         "back-pressure once the host runs ~2000 launches ahead"}}

    # The ONE exchange step of the queue — shared text-encoder / VAE weights from rank 0, scatter + all-gather per 1 GiB
    # bucket, every rank then encodes the same ids with ITS copy and the results are compared bit for bit — runs AFTER the
    # timed region and under a watchdog: N > 1 executes for the first time on the driver's node, and a stuck collective must
    # cost the weights report, not the measured line.  On expiry rank 0 prints the line with the failure recorded and every
    # rank leaves; an exception on a rank is gathered (MIN over an ok flag) so that no rank reports success alone.
    if distributed and not args.no_shared_weights:
        import threading
        mods, what, probe = shared_weights(args.workload, dev, rank)
        if mods is not None:
            def bail():
                if rank == 0:
                    _flush_c_stdio()
                    print(json.dumps(core_line({"weights": {"what": what, "verified": False,
                                                           "error": f"exchange step exceeded {args.exchange_timeout} s"}})),
                          flush=True)
                os._exit(0)
            timer = threading.Timer(args.exchange_timeout, bail)
            timer.daemon = True
            timer.start()
            err = None
            try:
                w = dict(render_queue.broadcast_parameters(mods, src=0), what=what)
                sums = probe(mods)                              # every rank encodes the same ids with ITS copy
                gathered = [torch.empty_like(sums) for _ in range(dist.get_world_size())]
                dist.all_gather(gathered, sums)
                w["verified"] = bool(all(torch.equal(g, gathered[0]) for g in gathered) and torch.isfinite(gathered[0]).all())
            except Exception as e:
                err = f"{type(e).__name__}: {e}"[:300]
                w = {"what": what, "error": err, "verified": False}
            ok = torch.tensor([0 if err else 1], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # reached by every rank that is not stuck; the watchdog covers the rest
            timer.cancel()
            if int(ok.item()) == 0 and not err:
                w = dict(w, verified=False, error="another rank failed in the exchange step")
            bcast["weights"] = w
            del mods
            torch.cuda.empty_cache()

    roofline = None
    kernels = {}
    if rank == 0 and not args.no_roofline:
        nprof = min(3, args.steps)
        reset(total)
        begin(0, nprof)
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
        # dominant kernel by time: the GEMM for flux / qwen, attention for wan
        dom = max(("gemm", "attention"), key=lambda k: prof[k]["ms"])
        gk = prof[dom]
        ach = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (tools/gpu_pmc*.sh), which cannot run
        # inside this process: the committed summary is used ONLY if it was taken from the kernel source this binary
        # was built from (sha256 recorded next to it); otherwise null — never a stale constant.
        suffix = {("gemm", "flux"): "pmc_gemm.json", ("gemm", "qwen"): "pmc_gemm_qwen.json",
                  ("attention", "wan"): "pmc_attn_wan.json"}.get((dom, args.workload))
        traffic, traffic_src, pmc_rec = pmc_traffic("gemm.hip" if dom == "gemm" else "attention.hip", suffix)
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel" if dom == "gemm" else "attn_fwd_d128_w64_kernel",
                    "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": 1e3 * gk["ms"] / gk["launches"],
                    "launches_per_step": gk["launches"] / nprof,
                    "algorithmic_tflop_per_step": gk["flops"] / nprof / 1e12,
                    "algorithmic_bytes_per_launch": gk["bytes"] / gk["launches"] if gk["bytes"] else None,
                    "traffic_over_algorithmic": (traffic * gk["launches"] / gk["bytes"]) if traffic and gk["bytes"] else None,
                    "note": "achieved = algorithmic flops / summed HIP-event durations of the kernel's launches over "
                            f"{nprof} extra event-instrumented steps; the per-class times in `kernels` come from that pass, "
                            "include the event overhead and are NOT additive to ms_per_step; `traffic` = HBM bytes PER LAUNCH "
                            "(mean over the kernel's launches) from the separate rocprofv3 --pmc pass named in traffic_source, "
                            "used only when that record's source hash equals this binary's kernel source; algorithmic_bytes_per_launch = "
                            "activation + weight + output bytes once (mean over the kernel's launches)"
                            + (f"; that pass ran `{pmc_rec.get('command')}`" + (" - a DEPTH-REDUCED run of the same launches: "
                               "per launch it is the full-depth figure, per step it is not" if "--layers" in str(pmc_rec.get("command")) else "")
                               if pmc_rec else "")}
        # The power roof the fraction has to be read against: the chip clocks to its power budget (DVFS), so the nominal
        # 2.5 PF (2.4 GHz) is not on offer while this kernel runs.  Measured live, un-instrumented steps, no profiler: every
        # GEMM workgroup stamps its K-loop with s_memtime (shader cycles) and s_memrealtime (100 MHz) — apexmi_clk_*.
        if dom == "gemm":
            nclk = min(5, args.steps)
            reset(total)                      # the scheduler's step index: the roofline pass above consumed some of the timesteps
            begin(0, nclk)
            lat = latents
            lib.clk_enable(True)
            for i in range(nclk):
                lat = step(i, lat)
            ck = lib.clk_read()
            lib.clk_enable(False)
            if ck["ghz"]:
                adj = PEAK_BF16_TFLOPS * ck["ghz"] / 2.4
                roofline.update({"clock_ghz": ck["ghz"], "clock_source": f"s_memtime / s_memrealtime over the GEMM K-loops of {nclk} "
                                 "un-instrumented steps", "peak_at_clock": adj, "frac_of_clock_adjusted_peak": ach / adj})

    clip = None
    if rank == 0 and clip_fn is not None and not args.no_clip and not args.layers:
        clip = clip_fn()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        full = not args.layers
        tf = STEP_TFLOP[args.workload]
        out = core_line()
        out.update({"sec_per_clip": clip["sec_per_clip"] if clip else None, "clip": clip, "roofline": roofline,
                    "kernels": kernels})
        if not args.no_cpu_baseline and args.gpus == 1 and args.workload in ("flux", "flux512", "qwen", "wan"):
            out["cpu_baseline"] = {"flux": cpu_baseline, "flux512": lambda: cpu_baseline(1024, 32), "qwen": cpu_baseline_qwen,
                                   "wan": cpu_baseline_wan}[args.workload]()
        if args.workload == "flux" and args.gpus == 1 and full and not args.no_wan:
            del step, latents, clip_fn
            torch.cuda.empty_cache()
            out["wan"] = wan_half(dev, cpu=not args.no_cpu_baseline)
    _flush_c_stdio()           # RCCL's version banner sits in C stdio: get it out BEFORE the result line
    if rank == 0:
        print(json.dumps(out), flush=True)    # the ONE JSON line, last on stdout
    if distributed:
        # The line is out: a rank that never reaches this barrier (or a teardown that hangs) must not hold the job open —
        # leave after a minute whatever happens.
        import threading
        t = threading.Timer(60.0, lambda: os._exit(0))
        t.daemon = True
        t.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001  (teardown only; the result is already printed)
            print(f"[bench] teardown: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        t.cancel()


if __name__ == "__main__":
    main()
