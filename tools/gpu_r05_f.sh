#!/bin/bash
# the default bench line with the thread-swept CPU baselines, then BASELINE config 0 in full on the host cores (minutes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
( time timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05f/bench_default.json 2> gpurun_out/r05f/bench_default.err ) 2> gpurun_out/r05f/bench_default.time
tail -1 gpurun_out/r05f/bench_default.json | cut -c1-300; tail -3 gpurun_out/r05f/bench_default.time
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "code_point" > gpurun_out/r05f/tests.log 2>&1; grep -E "passed|failed|^\[" gpurun_out/r05f/tests.log | tail -8
timeout 3300 python tools/cpu_flux512_full.py > gpurun_out/r05f/cpu_flux512_full.json 2> gpurun_out/r05f/cpu_flux512_full.err
tail -1 gpurun_out/r05f/cpu_flux512_full.json | cut -c1-600; tail -5 gpurun_out/r05f/cpu_flux512_full.err
