#!/bin/bash
# round 4, visit C: the free-running ring GEMM schedule (gemm.large / gemm.config = 9) against the shipped ping-pong (7): bit-identity on
# ragged / gated / K = 64 problems, per-shape TFLOP/s, the step's sustained GEMM sequence, and the whole Flux step with live clocks.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KEY=gemm.config VALS=7,9,10 timeout 900 python tools/gemm_key_ab.py > gpurun_out/r04_gemm_ring_ab.log 2> gpurun_out/r04_c.err; cat gpurun_out/r04_gemm_ring_ab.log; tail -3 gpurun_out/r04_c.err
timeout 600 python tools/gemm_seq_bench.py > gpurun_out/r04_gemm_ring_seq.log 2>> gpurun_out/r04_c.err; cat gpurun_out/r04_gemm_ring_seq.log
CLK=1 ARMS="base;gemm.large=9;gemm.large=10" STEPS=14 ROUNDS=3 timeout 700 python tools/flux_step_ab.py > gpurun_out/r04_ab_ring_step.log 2>> gpurun_out/r04_c.err; tail -2 gpurun_out/r04_ab_ring_step.log; tail -3 gpurun_out/r04_c.err
