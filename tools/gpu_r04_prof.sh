#!/bin/bash
# Round-4 evidence run: hash-matched PMC records of the GEMM (Flux, Qwen) and of the Wan attention, rocprofv3 kernel stats of the
# default Flux command, then the driver-style bench lines (which pick the PMC records up by the kernel source's sha256).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/gpu_pmc_gemm.sh > gpurun_out/r04/pmc_gemm_flux.log 2>&1; tail -3 gpurun_out/r04/pmc_gemm_flux.log
WORKLOAD=qwen bash tools/gpu_pmc_gemm.sh > gpurun_out/r04/pmc_gemm_qwen.log 2>&1; tail -3 gpurun_out/r04/pmc_gemm_qwen.log
cp gpurun_out/pmc_gemm_flux/r04_pmc_gemm.json gpurun_out/pmc_gemm_qwen/r04_pmc_gemm_qwen.json profiles/ 2>/dev/null
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r04flux > gpurun_out/r04/profile_flux.log 2>&1; tail -3 gpurun_out/r04/profile_flux.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_default_b.json 2> gpurun_out/r04/bench_default_b.err; tail -1 gpurun_out/r04/bench_default_b.json | cut -c1-300
timeout 900 python bench.py --workload qwen --steps 8 --warmup 2 > gpurun_out/r04/bench_qwen_b.json 2> gpurun_out/r04/bench_qwen_b.err; tail -1 gpurun_out/r04/bench_qwen_b.json | cut -c1-200
timeout 900 python bench.py --workload qwen --steps 8 --warmup 2 --no-mod-table --no-cpu-baseline --no-roofline > gpurun_out/r04/bench_qwen_nomod.json 2> /dev/null; tail -1 gpurun_out/r04/bench_qwen_nomod.json | cut -c1-200
