#!/bin/bash
# round 4, visit B: the GEMM tile's stall-free stream (tools/ubench/gemm_roof) and the live clock of the shipped ping-pong GEMM vs the
# one-wave-per-SIMD form (gemm.large=6) inside the real step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/ubench/gemm_roof 8 > gpurun_out/r04_gemm_roof.log 2>&1; cat gpurun_out/r04_gemm_roof.log
CLK=1 ARMS="attr:fuse_qkv=0;attr:fuse_qkv=0,gemm.large=6;base" STEPS=14 ROUNDS=3 timeout 700 python tools/flux_step_ab.py > gpurun_out/r04_ab_clock_sched.log 2> gpurun_out/r04_ab.err; tail -2 gpurun_out/r04_ab_clock_sched.log; tail -3 gpurun_out/r04_ab.err
