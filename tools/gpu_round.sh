#!/bin/bash
# One GPU-box visit: parity tests (crash-isolated via xdist), smoke, bench, rocprofv3 kernel stats.
# Usage (through gpurun): bash tools/gpu_round.sh [quick]
set -u
OUT=gpurun_out
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocm-smi ==" > $OUT/env.log; rocm-smi --showproductname 2>&1 | head -20 >> $OUT/env.log
nproc >> $OUT/env.log; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/env.log
python - <<'PY' >> $OUT/env.log 2>&1
import torch
print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
p = torch.cuda.get_device_properties(0)
print(p.multi_processor_count, p.total_memory/2**30, "GiB")
PY
echo "== pytest gpu ==" 
timeout 1500 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 30 -x --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu_x.log 2>&1
echo "pytest -x exit: $?" | tee -a $OUT/summary.log
timeout 1800 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 30 --no-header -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee -a $OUT/summary.log
tail -60 $OUT/pytest_gpu.log
echo "== smoke =="
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit: $?" | tee -a $OUT/summary.log
tail -5 $OUT/smoke.log
echo "== bench reduced =="
timeout 900 python bench.py --layers 2,2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_small.log 2>&1; echo "bench small exit: $?" | tee -a $OUT/summary.log
tail -3 $OUT/bench_small.log
if [ "${1:-}" != "quick" ]; then
echo "== bench full =="
timeout 1500 python bench.py --steps 10 --warmup 3 > $OUT/bench_full.log 2>&1; echo "bench full exit: $?" | tee -a $OUT/summary.log
tail -3 $OUT/bench_full.log
echo "== rocprof =="
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o flux -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
echo "rocprof exit: $?" | tee -a $GRAFT_REPO_ROOT/$OUT/summary.log
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -30 $f; done
# keep only the small csv summaries (traces can be large)
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
fi
