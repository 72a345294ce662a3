#!/bin/bash
# A/B on one box: Flux step with / without the MLP-up GEMM overlapped under attention (interleaved runs)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in 0 1; do
    APEX_FLUX_OVERLAP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clip --no-wan 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('overlap=$v', round(d['ms_per_step'],2), 'ms/step  gemm', round(d['kernels']['gemm']['ms_per_step'],2), 'attn', round(d['kernels']['attention']['ms_per_step'],2), 'finite', d['finite'])"
  done
done
