#!/bin/bash
# Round-3 checkpoint: full GPU suite + Qwen / Hunyuan A/B of the fused QKV epilogue + default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_gpu.log 2>&1; tail -5 gpurun_out/r03/pytest_gpu.log
for r in 1 2; do for f in 0 1; do
  APEX_FUSE_QKV=$f timeout 400 python bench.py --workload qwen --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': 'qwen', 'fuse_qkv': $f, 'ms_per_step': d['ms_per_step']}))"
done; done | tee gpurun_out/r03/ab_fuse_qwen.log
for f in 0 1; do
  APEX_FUSE_QKV=$f timeout 400 python bench.py --workload hunyuan --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': 'hunyuan', 'fuse_qkv': $f, 'ms_per_step': d['ms_per_step']}))"
done | tee gpurun_out/r03/ab_fuse_hunyuan.log
timeout 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -1 gpurun_out/r03/bench_default.json | cut -c1-600
