#!/bin/bash
# The Qwen text-stream tail GEMMs: 4-wave (gemm.config 1) vs 8-wave (8) 128x128 tiling vs the 256x256 tiling (7), cold weights; then
# the Qwen step with gemm.tail = 1 / 2, interleaved.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
GEMM_CFGS=1,8,7 GEMM_COLD=1 GEMM_SHAPES=qwen_ff_up_txt,qwen_ff_down_txt,qkv_txt timeout 300 python tools/gemm_bench.py 2>&1 | tail -6 | tee gpurun_out/r03/tail_gemm_bench.log
timeout 200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for r in 1 2; do for t in 1 2; do
  timeout 400 python bench.py --workload qwen --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --tune gemm.tail=$t 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': 'qwen', 'gemm.tail': $t, 'ms_per_step': d['ms_per_step']}))"
done; done | tee gpurun_out/r03/ab_tail_qwen.log
