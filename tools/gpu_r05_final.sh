#!/bin/bash
# Round-5 final evidence with the shipped binary: the whole GPU suite + smoke, hash-matched PMC records (GEMM flux / qwen), rocprofv3
# kernel stats of the default Flux command, then the driver-style bench lines (flux default incl. the Wan half, flux512, qwen, wan, hunyuan).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z
( time timeout 2400 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider > gpurun_out/r05z/suite.log 2>&1 ) 2> gpurun_out/r05z/suite.time; echo "suite rc $?"; tail -3 gpurun_out/r05z/suite.log; tail -3 gpurun_out/r05z/suite.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z/smoke.log 2>&1; tail -3 gpurun_out/r05z/smoke.log
ROUND=r05 bash tools/gpu_pmc_gemm.sh > gpurun_out/r05z/pmc_gemm_flux.log 2>&1; tail -2 gpurun_out/r05z/pmc_gemm_flux.log
ROUND=r05 WORKLOAD=qwen bash tools/gpu_pmc_gemm.sh > gpurun_out/r05z/pmc_gemm_qwen.log 2>&1; tail -2 gpurun_out/r05z/pmc_gemm_qwen.log
cp gpurun_out/pmc_gemm_flux/r05_pmc_gemm.json gpurun_out/pmc_gemm_qwen/r05_pmc_gemm_qwen.json profiles/ 2>/dev/null
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r05flux > gpurun_out/r05z/profile_flux.log 2>&1; tail -2 gpurun_out/r05z/profile_flux.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05z/bench_default.json 2> gpurun_out/r05z/bench_default.err; tail -1 gpurun_out/r05z/bench_default.json | cut -c1-300
timeout 600 python bench.py --workload flux512 --steps 30 --warmup 5 > gpurun_out/r05z/bench_flux512.json 2> gpurun_out/r05z/bench_flux512.err; tail -1 gpurun_out/r05z/bench_flux512.json | cut -c1-200
timeout 900 python bench.py --workload qwen --steps 8 --warmup 2 > gpurun_out/r05z/bench_qwen.json 2> gpurun_out/r05z/bench_qwen.err; tail -1 gpurun_out/r05z/bench_qwen.json | cut -c1-200
timeout 900 python bench.py --workload wan --steps 2 --warmup 1 > gpurun_out/r05z/bench_wan.json 2> gpurun_out/r05z/bench_wan.err; tail -1 gpurun_out/r05z/bench_wan.json | cut -c1-200
timeout 900 python bench.py --workload hunyuan --steps 3 --warmup 1 > gpurun_out/r05z/bench_hunyuan.json 2> gpurun_out/r05z/bench_hunyuan.err; tail -1 gpurun_out/r05z/bench_hunyuan.json | cut -c1-200
for w in wan hunyuan flux; do timeout 400 python tools/vae_bench.py $w 3 2>&1 | tail -1; done | tee gpurun_out/r05z/vae_bench.log
