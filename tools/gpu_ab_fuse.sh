#!/bin/bash
# Interleaved A/B of the Flux step with the q/k/v preparation fused into the QKV GEMM's epilogue (APEX_FLUX_FUSE_QKV=1) and as a
# separate pass (=0), same box, alternating runs.  Writes gpurun_out/ab_fuse.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/ab_fuse.log
for r in 1 2 3; do
  for f in 0 1; do
    APEX_FLUX_FUSE_QKV=$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-clip --no-wan 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'fuse_qkv': $f, 'round': $r, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" >> gpurun_out/ab_fuse.log
  done
done
cat gpurun_out/ab_fuse.log
