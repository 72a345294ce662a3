#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "x384" > gpurun_out/r05g/tests.log 2>&1
echo "tests rc $?"; tail -5 gpurun_out/r05g/tests.log
KEY=gemm.x384 SHAPES="4608,21504,3072,gelu;4608,12288,3072,gelu;4608,9216,3072,bias;4608,3072,15360,gate_res;4608,3072,3072,gate_res;8448,12288,3072,gelu;75648,5120,5120,bias;75648,13824,5120,gelu;75648,5120,13824,gate_res" ROUNDS=3 REPS=24 timeout 900 python tools/gemm_x288_ab.py > gpurun_out/r05g/gemm_x384_ab.log 2>&1; cat gpurun_out/r05g/gemm_x384_ab.log | cut -c1-420
