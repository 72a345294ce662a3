#!/bin/bash
# Round-3 evidence: GEMM group_m A/B, rocprofv3 kernel stats + PMC of the Flux and Qwen steps, PMC records for roofline.traffic.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "key_padding" 2>&1 | tail -2
: > gpurun_out/r03/ab_group_m.log
for r in 1 2; do for gm in 8 4 6 16; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-clip --no-wan --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'gemm.group_m': $gm, 'round': $r, 'ms_per_step': d['ms_per_step']}))" >> gpurun_out/r03/ab_group_m.log
done; done
cat gpurun_out/r03/ab_group_m.log
bash tools/gpu_pmc_gemm.sh 2>&1 | tail -25
WORKLOAD=qwen bash tools/gpu_pmc_gemm.sh 2>&1 | tail -25
bash tools/gpu_profile.sh r03_flux1024 2>&1 | tail -4
bash tools/gpu_profile.sh r03_qwen python $GRAFT_REPO_ROOT/bench.py --workload qwen --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -4
