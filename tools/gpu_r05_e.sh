#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "code_point or small_gate or x288" > gpurun_out/r05e/tests.log 2>&1
echo "tests rc $?"; grep -E "passed|failed|rror|^\[" gpurun_out/r05e/tests.log | tail -14
timeout 600 python bench.py --workload flux512 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05e/bench_flux512.json 2> gpurun_out/r05e/bench_flux512.err; tail -1 gpurun_out/r05e/bench_flux512.json | cut -c1-700
ARMS="gemm.small_max=0;base" STEPS=12 ROUNDS=3 timeout 600 python tools/flux_step_ab.py > gpurun_out/r05e/flux_step_ab_small.log 2>&1; tail -1 gpurun_out/r05e/flux_step_ab_small.log
