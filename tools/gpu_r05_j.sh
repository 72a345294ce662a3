#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_gemm_qkv.py tests/test_gpu_ops.py -m gpu -x -q -k "384 or qkv or fus or joint or single_block" > gpurun_out/r05j/tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r05j/tests.log
ARMS="gemm.x384=0;base;gemm.x384_qkv=0" STEPS=12 ROUNDS=4 CLK=1 timeout 900 python tools/flux_step_ab.py > gpurun_out/r05j/flux_step_ab_x384.log 2>&1; tail -1 gpurun_out/r05j/flux_step_ab_x384.log
