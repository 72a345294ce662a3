#!/bin/bash
# Round 5, second call: the 288 x 192 exact-fill GEMM tiling — tests, per-shape A/B, step A/B (flux 1024^2), qwen line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "x288" > gpurun_out/r05b/tests.log 2>&1
echo "tests rc $?"; tail -5 gpurun_out/r05b/tests.log
timeout 900 python tools/gemm_x288_ab.py > gpurun_out/r05b/gemm_x288_ab.log 2>&1; cat gpurun_out/r05b/gemm_x288_ab.log | cut -c1-400
ARMS="gemm.x288=0;base" STEPS=12 ROUNDS=4 CLK=1 timeout 900 python tools/flux_step_ab.py > gpurun_out/r05b/flux_step_ab.log 2>&1; tail -1 gpurun_out/r05b/flux_step_ab.log
timeout 600 python bench.py --workload qwen --steps 8 --warmup 2 --no-cpu-baseline --tune gemm.x288=0 > gpurun_out/r05b/bench_qwen_x0.json 2> gpurun_out/r05b/bench_qwen_x0.err; tail -1 gpurun_out/r05b/bench_qwen_x0.json | cut -c1-200
timeout 600 python bench.py --workload qwen --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r05b/bench_qwen_x1.json 2> gpurun_out/r05b/bench_qwen_x1.err; tail -1 gpurun_out/r05b/bench_qwen_x1.json | cut -c1-200
timeout 600 python bench.py --workload flux512 --steps 30 --warmup 5 --no-cpu-baseline --no-clip > gpurun_out/r05b/bench_flux512.json 2> gpurun_out/r05b/bench_flux512.err; tail -1 gpurun_out/r05b/bench_flux512.json | cut -c1-1200
