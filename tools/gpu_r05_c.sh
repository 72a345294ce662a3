#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 600 python tools/gemm_fill_probe.py > gpurun_out/r05c/fill_probe_256.log 2>&1; cat gpurun_out/r05c/fill_probe_256.log | grep tile
X288=1 timeout 600 python tools/gemm_fill_probe.py > gpurun_out/r05c/fill_probe_288.log 2>&1; cat gpurun_out/r05c/fill_probe_288.log | grep tile
ARMS="base;gemm.large=6" STEPS=12 ROUNDS=3 CLK=1 timeout 600 python tools/flux_step_ab.py > gpurun_out/r05c/flux_step_ab_cfg6.log 2>&1; tail -1 gpurun_out/r05c/flux_step_ab_cfg6.log
