#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_weights.py tests/test_gpu_ops.py tests/test_lora.py -m gpu -x -q -s -k "lora or x288 or fp8" > gpurun_out/r05d/tests.log 2>&1
echo "tests rc $?"; grep -E "passed|failed|rror|\[fp8" gpurun_out/r05d/tests.log | tail -12
timeout 300 tools/ubench/gemm_roof 4 > gpurun_out/r05d/gemm_roof.log 2>&1; cat gpurun_out/r05d/gemm_roof.log | cut -c1-260
timeout 600 python tools/gemm_small_ab.py > gpurun_out/r05d/gemm_small_ab.log 2>&1; grep shape gpurun_out/r05d/gemm_small_ab.log
