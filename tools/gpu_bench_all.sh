#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_flux.log 2>&1; echo "flux exit $?"; tail -1 $OUT/bench_flux.log | cut -c1-1200
timeout 900 python bench.py --workload qwen --steps 5 --warmup 2 > $OUT/bench_qwen.log 2>&1; echo "qwen exit $?"; tail -1 $OUT/bench_qwen.log | cut -c1-700
timeout 1500 python bench.py --workload wan --steps 2 --warmup 1 ${WAN_ARGS:-} > $OUT/bench_wan.log 2>&1; echo "wan exit $?"; tail -1 $OUT/bench_wan.log | cut -c1-900
