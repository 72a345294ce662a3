#!/bin/bash
# end of round 4, final binary: full GPU suite, smoke, the Wan VAE tile under rocprofv3 / PMC, decodes, default / qwen / wan bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout 2800 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04h/gpu_suite_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04h/smoke.txt
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r04_vae_wan_tile python $GRAFT_REPO_ROOT/tools/vae_bench.py wan-tile 1 > gpurun_out/r04h/profile_vae.log 2>&1; tail -2 gpurun_out/r04h/profile_vae.log
for w in wan hunyuan flux taehv; do timeout 400 python tools/vae_bench.py $w 3 2>&1 | tail -1; done | tee gpurun_out/r04h/vae_bench.log
timeout 900 python bench.py > gpurun_out/r04h/bench_default.json 2> gpurun_out/r04h/bench_default.err; tail -1 gpurun_out/r04h/bench_default.json | cut -c1-200
