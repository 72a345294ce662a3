#!/bin/bash
# round 5: attn.w64 as the shipped main launch — attention tests, then step-level A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05o
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_like_for_like.py -m gpu -q -x -k "attn or attention or sdpa or tail_split" 2>&1 | tail -4
for w in 1 0 1 0; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-clip --no-wan --tune attn.w64=$w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flux w64=$w', round(d['ms_per_step'],3), {k:round(v.get('ms_per_step',0),3) if isinstance(v,dict) else v for k,v in d.get('kernels',{}).items()})"; done 2>&1 | tee gpurun_out/r05o/flux_step_ab.log
