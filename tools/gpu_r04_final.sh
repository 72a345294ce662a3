#!/bin/bash
# Round-4 final evidence with the shipped binary: hash-matched PMC records (GEMM flux / qwen, Wan attention), rocprofv3 kernel stats of the
# default Flux command and of one Wan VAE tile, then the driver-style bench lines (flux default incl. the Wan half, qwen, wan).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
bash tools/gpu_pmc_gemm.sh > gpurun_out/r04f/pmc_gemm_flux.log 2>&1; tail -2 gpurun_out/r04f/pmc_gemm_flux.log
WORKLOAD=qwen bash tools/gpu_pmc_gemm.sh > gpurun_out/r04f/pmc_gemm_qwen.log 2>&1; tail -2 gpurun_out/r04f/pmc_gemm_qwen.log
bash tools/gpu_pmc_wan.sh > gpurun_out/r04f/pmc_wan.log 2>&1; tail -2 gpurun_out/r04f/pmc_wan.log
cp gpurun_out/pmc_gemm_flux/r04_pmc_gemm.json gpurun_out/pmc_gemm_qwen/r04_pmc_gemm_qwen.json gpurun_out/pmc_wan/r04_pmc_attn_wan.json profiles/ 2>/dev/null
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r04flux > gpurun_out/r04f/profile_flux.log 2>&1; tail -2 gpurun_out/r04f/profile_flux.log
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r04_vae_wan_tile python $GRAFT_REPO_ROOT/tools/vae_bench.py wan-tile 1 > gpurun_out/r04f/profile_vae.log 2>&1; tail -2 gpurun_out/r04f/profile_vae.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err; tail -1 gpurun_out/r04f/bench_default.json | cut -c1-300
timeout 900 python bench.py --workload qwen --steps 8 --warmup 2 > gpurun_out/r04f/bench_qwen.json 2> gpurun_out/r04f/bench_qwen.err; tail -1 gpurun_out/r04f/bench_qwen.json | cut -c1-200
timeout 900 python bench.py --workload wan --steps 2 --warmup 1 > gpurun_out/r04f/bench_wan.json 2> gpurun_out/r04f/bench_wan.err; tail -1 gpurun_out/r04f/bench_wan.json | cut -c1-200
for w in wan hunyuan flux; do timeout 400 python tools/vae_bench.py $w 3 2>&1 | tail -1; done | tee gpurun_out/r04f/vae_bench.log
