#!/bin/bash
# Round-4 checkpoint: full GPU suite + smoke + default bench line (Flux + Wan half) + the qwen bench.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04/pytest_gpu.log 2>&1; tail -8 gpurun_out/r04/pytest_gpu.log; grep -E "^(FAILED|ERROR)" gpurun_out/r04/pytest_gpu.log | head -20
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r04/smoke.log 2>&1; tail -2 gpurun_out/r04/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; tail -1 gpurun_out/r04/bench_default.json | cut -c1-400
timeout 600 python bench.py --workload qwen --steps 8 --warmup 2 > gpurun_out/r04/bench_qwen.json 2> gpurun_out/r04/bench_qwen.err; tail -1 gpurun_out/r04/bench_qwen.json | cut -c1-300
