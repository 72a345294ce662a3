#!/bin/bash
# Round-3 closing run: GPU suite, hash-matched PMC records, the driver-style bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/r03/pytest_gpu_final.log
bash tools/gpu_pmc_gemm.sh > gpurun_out/r03/pmc_gemm_flux.log 2>&1; tail -3 gpurun_out/r03/pmc_gemm_flux.log
WORKLOAD=qwen bash tools/gpu_pmc_gemm.sh > gpurun_out/r03/pmc_gemm_qwen.log 2>&1; tail -3 gpurun_out/r03/pmc_gemm_qwen.log
bash tools/gpu_pmc_wan.sh > gpurun_out/r03/pmc_attn_wan.log 2>&1; tail -3 gpurun_out/r03/pmc_attn_wan.log

cp gpurun_out/pmc_gemm_flux/r03_pmc_gemm.json gpurun_out/pmc_gemm_qwen/r03_pmc_gemm_qwen.json profiles/ 2>/dev/null
cp gpurun_out/pmc_wan/r03_pmc_attn_wan.json profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -1 gpurun_out/r03/bench_default.json | cut -c1-300
timeout 900 python bench.py --workload qwen > gpurun_out/r03/bench_qwen.json 2> gpurun_out/r03/bench_qwen.err; tail -1 gpurun_out/r03/bench_qwen.json | cut -c1-200
timeout 1500 python bench.py --workload wan --steps 2 --warmup 1 > gpurun_out/r03/bench_wan.json 2> gpurun_out/r03/bench_wan.err; tail -1 gpurun_out/r03/bench_wan.json | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
