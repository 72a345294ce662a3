#!/bin/bash
# same-box A/B of the shipped attention kernel in the steps, and the whole Wan clip
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05q
for w in 1 0 1 0; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-clip --no-wan --tune attn.w64=$w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flux w64=$w', round(d['ms_per_step'],3), {k:round(v.get('ms_per_step',0),3) if isinstance(v,dict) else v for k,v in d.get('kernels',{}).items()})"; done 2>&1 | tee gpurun_out/r05q/flux_step_ab.log
for w in 1 0; do timeout 600 python bench.py --workload wan --steps 2 --warmup 1 --no-cpu-baseline --no-clip --tune attn.w64=$w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wan w64=$w', round(d['ms_per_step'],1), d.get('roofline',{}).get('achieved'), {k:round(v.get('ms_per_step',0),1) if isinstance(v,dict) else v for k,v in d.get('kernels',{}).items()})"; done 2>&1 | tee gpurun_out/r05q/wan_step_ab.log
timeout 900 python bench.py --workload wan --clip --no-cpu-baseline > gpurun_out/r05q/bench_wan_clip.json 2> gpurun_out/r05q/bench_wan_clip.err; tail -1 gpurun_out/r05q/bench_wan_clip.json | cut -c1-400
