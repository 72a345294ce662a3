#!/bin/bash
# gemm.group_m 8 vs 6 on the three step workloads, interleaved, same box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
: > gpurun_out/r03/ab_group_m_all.log
for r in 1 2; do for gm in 8 6; do
  for w in flux qwen; do
    timeout 400 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-clip --no-wan --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': '$w', 'gemm.group_m': $gm, 'round': $r, 'ms_per_step': round(d['ms_per_step'], 3)}))" >> gpurun_out/r03/ab_group_m_all.log
  done
  timeout 600 python bench.py --workload wan --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': 'wan', 'gemm.group_m': $gm, 'round': $r, 'ms_per_step': round(d['ms_per_step'], 1)}))" >> gpurun_out/r03/ab_group_m_all.log
done; done
cat gpurun_out/r03/ab_group_m_all.log
