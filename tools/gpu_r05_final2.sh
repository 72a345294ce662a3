#!/bin/bash
# Round-5 final evidence, second session (binary with the w64 attention kernel): hash-matched PMC record of the attention kernels in
# the Wan step, rocprofv3 kernel stats + PMC summary of the default Flux command, then the driver-style bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z2
ROUND=r05 PROF_TIMEOUT=900 bash tools/gpu_pmc_wan.sh > gpurun_out/r05z2/pmc_attn_wan.log 2>&1; tail -3 gpurun_out/r05z2/pmc_attn_wan.log | cut -c1-200
cp gpurun_out/pmc_wan/r05_pmc_attn_wan.json profiles/ 2>/dev/null
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r05flux2 > gpurun_out/r05z2/profile_flux.log 2>&1; tail -2 gpurun_out/r05z2/profile_flux.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05z2/bench_default.json 2> gpurun_out/r05z2/bench_default.err; tail -1 gpurun_out/r05z2/bench_default.json | cut -c1-300
timeout 600 python bench.py --workload flux512 --steps 30 --warmup 5 > gpurun_out/r05z2/bench_flux512.json 2> gpurun_out/r05z2/bench_flux512.err; tail -1 gpurun_out/r05z2/bench_flux512.json | cut -c1-200
timeout 900 python bench.py --workload qwen --steps 8 --warmup 2 > gpurun_out/r05z2/bench_qwen.json 2> gpurun_out/r05z2/bench_qwen.err; tail -1 gpurun_out/r05z2/bench_qwen.json | cut -c1-200
timeout 900 python bench.py --workload wan --steps 2 --warmup 1 > gpurun_out/r05z2/bench_wan.json 2> gpurun_out/r05z2/bench_wan.err; tail -1 gpurun_out/r05z2/bench_wan.json | cut -c1-200
timeout 900 python bench.py --workload hunyuan --steps 3 --warmup 1 > gpurun_out/r05z2/bench_hunyuan.json 2> gpurun_out/r05z2/bench_hunyuan.err; tail -1 gpurun_out/r05z2/bench_hunyuan.json | cut -c1-200
