#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wan.py -m gpu -x -q -k "x384 or x288 or wan" > gpurun_out/r05i/tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r05i/tests.log
for x in 0 1 0 1; do timeout 600 python bench.py --workload wan --steps 2 --warmup 1 --no-cpu-baseline --no-clip --tune gemm.x384=$x > gpurun_out/r05i/bench_wan_x$x.json 2> gpurun_out/r05i/bench_wan_x$x.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05i/bench_wan_x$x.json").read().strip().splitlines()[-1])
print("x384=$x", round(d["ms_per_step"],1), {k:(round(v["ms_per_step"],1), round(v["tflops"] or 0)) for k,v in d["kernels"].items() if k in ("gemm","attention")})
PY
done
