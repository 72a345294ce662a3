#!/bin/bash
# round 4, visit A: new tests (multi-row GEMV, modulation table, inference mode), ln_modulate microbench, in-process A/B of the
# Flux step (modulation table on / off, ln.wave 1 / 2), default bench line with the live clock.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_test.sh tests/test_gpu_ops.py tests/test_gpu_flux.py tests/test_gpu_end_to_end.py -k "gemv or ln_modulate or flux"
timeout 300 python tools/ln_bench.py > gpurun_out/r04_ln_bench.json 2> gpurun_out/r04_ln_bench.err; cat gpurun_out/r04_ln_bench.json
ARMS="base;modtable=0;ln.wave=2;modtable=0,ln.wave=2" STEPS=14 ROUNDS=3 timeout 600 python tools/flux_step_ab.py > gpurun_out/r04_ab_modtable_ln.log 2> gpurun_out/r04_ab.err; tail -2 gpurun_out/r04_ab_modtable_ln.log; tail -3 gpurun_out/r04_ab.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-wan > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; tail -c 2500 gpurun_out/r04_bench_a.json; tail -3 gpurun_out/r04_bench_a.err
