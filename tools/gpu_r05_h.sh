#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "x384" > gpurun_out/r05h/tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r05h/tests.log
S="4608,21504,3072,gelu;4608,9216,3072,bias;75648,13824,5120,gelu;75648,5120,13824,gate_res"
for d in 0 1 0 1; do KEY=gemm.x384 TUNE=gemm.x384_dist=$d SHAPES="$S" ROUNDS=3 REPS=24 timeout 600 python tools/gemm_x288_ab.py 2>&1 | grep shape | sed "s/^/dist$d /" | cut -c1-260; done | tee gpurun_out/r05h/x384_dist_ab.log
X384=1 DIST=1 timeout 600 python tools/gemm_fill_probe.py 2>&1 | grep tile | tee gpurun_out/r05h/fill_probe_384.log
