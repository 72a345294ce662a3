#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05o
for w in 1 0 1 0; do timeout 600 python bench.py --workload wan --steps 2 --warmup 1 --no-cpu-baseline --no-clip --tune attn.w64=$w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wan w64=$w', round(d['ms_per_step'],1), d.get('roofline',{}).get('achieved'), {k:round(v.get('ms_per_step',0),1) if isinstance(v,dict) else v for k,v in d.get('kernels',{}).items()})"; done 2>&1 | tee gpurun_out/r05o/wan_step_ab.log
