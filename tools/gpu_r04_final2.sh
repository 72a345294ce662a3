#!/bin/bash
# after the convolution-loop rewrite: full GPU suite, smoke, the Wan VAE tile under rocprofv3 / PMC, decodes, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 2800 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04g/gpu_suite_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04g/smoke.txt
PROF_TIMEOUT=600 bash tools/gpu_profile.sh r04_vae_wan_tile python $GRAFT_REPO_ROOT/tools/vae_bench.py wan-tile 1 > gpurun_out/r04g/profile_vae.log 2>&1; tail -2 gpurun_out/r04g/profile_vae.log
for w in wan hunyuan flux taehv; do timeout 400 python tools/vae_bench.py $w 3 2>&1 | tail -1; done | tee gpurun_out/r04g/vae_bench.log
timeout 900 python bench.py > gpurun_out/r04g/bench_default.json 2> gpurun_out/r04g/bench_default.err; tail -1 gpurun_out/r04g/bench_default.json | cut -c1-200
