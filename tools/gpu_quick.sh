#!/bin/bash
# Short GPU visit: parity tests + kernel A/B microbench + full bench (no rocprof).
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -n 1 --max-worker-restart 30 --no-header -rA -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee $OUT/summary.log
grep -E "^(PASSED|FAILED|ERROR)|passed|failed|hip vs|rel " $OUT/pytest_gpu.log | grep -v PASSED | tail -40
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit: $?" | tee -a $OUT/summary.log; tail -2 $OUT/smoke.log
timeout 900 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1; echo "gemm_bench exit: $?" | tee -a $OUT/summary.log; cat $OUT/gemm_bench.log | grep -v amdgpu.ids
timeout 900 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS:-} > $OUT/bench_full.log 2>&1; echo "bench exit: $?" | tee -a $OUT/summary.log; tail -1 $OUT/bench_full.log
