#!/bin/bash
# Round 5, first call: the new GPU tests (Wan full depth, f32-mode vs reference-run goldens, per-clip schedule handles), then the
# bench lines with host_enqueue_ms_per_step: default (flux 1024^2 + the Wan half), flux512, wan bf16 vs wan --fp8.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_f32_storage.py tests/test_gpu_flux.py tests/test_gpu_qwen.py -m gpu -x -q -s \
  -k "full_depth_two_experts or reference_run_golden or schedul or two_clips or wan_block_at_75600" > gpurun_out/r05a/tests.log 2>&1
echo "tests rc $?"; grep -E "passed|failed|error|\[full depth\]|\[f32-storage vs|\[full length\]" gpurun_out/r05a/tests.log | tail -20
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; tail -1 gpurun_out/r05a/bench_default.json | cut -c1-400
timeout 600 python bench.py --workload flux512 --steps 30 --warmup 5 > gpurun_out/r05a/bench_flux512.json 2> gpurun_out/r05a/bench_flux512.err; tail -1 gpurun_out/r05a/bench_flux512.json | cut -c1-900
timeout 900 python bench.py --workload wan --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05a/bench_wan.json 2> gpurun_out/r05a/bench_wan.err; tail -1 gpurun_out/r05a/bench_wan.json | cut -c1-300
timeout 900 python bench.py --workload wan --fp8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05a/bench_wan_fp8.json 2> gpurun_out/r05a/bench_wan_fp8.err; tail -1 gpurun_out/r05a/bench_wan_fp8.json | cut -c1-300
tail -3 gpurun_out/r05a/*.err
