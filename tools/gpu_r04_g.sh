#!/bin/bash
# group_m re-tune after the epilogue change: interleaved arms on the Flux step, then Qwen / Wan bench lines per value
mkdir -p gpurun_out; : > gpurun_out/r04_ab_group_m_c.log
ARMS="gemm.group_m=6;gemm.group_m=3;gemm.group_m=4;gemm.group_m=2;gemm.group_m=6;gemm.group_m=3;gemm.group_m=4;gemm.group_m=2" STEPS=12 ROUNDS=3 timeout 1200 python tools/flux_step_ab.py 2>&1 | grep '"arm"' | python -c "
import sys,json,collections,statistics
r=collections.defaultdict(list)
for l in sys.stdin:
    d=json.loads(l); r[d['arm']].append(d['ms_per_step'])
print(json.dumps({'flux_ms_per_step_median':{k:round(statistics.median(v),2) for k,v in r.items()},'n':{k:len(v) for k,v in r.items()}}))" >> gpurun_out/r04_ab_group_m_c.log
for rep in 1 2; do for gm in 6 3 4; do
  echo "qwen group_m=$gm $(timeout 600 python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],2))')" >> gpurun_out/r04_ab_group_m_c.log
done; done
for gm in 6 3 6 3; do
  echo "wan group_m=$gm $(timeout 600 python bench.py --workload wan --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],1))')" >> gpurun_out/r04_ab_group_m_c.log
done
cat gpurun_out/r04_ab_group_m_c.log
