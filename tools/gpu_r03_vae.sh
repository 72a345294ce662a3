set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03vae
timeout 300 python tools/conv_pp_bench.py > gpurun_out/r03vae/conv_pp_bench.log 2>&1
timeout 200 python tools/conv_ablate.py > gpurun_out/r03vae/conv_ablate.log 2>&1
timeout 200 python tools/conv_prof.py > gpurun_out/r03vae/conv_prof.log 2>&1
for pp in 0 1 2; do TUNE=conv.pp=$pp timeout 300 python tools/vae_bench.py wan 3 > gpurun_out/r03vae/wan_pp$pp.log 2>&1; tail -2 gpurun_out/r03vae/wan_pp$pp.log; done
for pp in 0 1; do TUNE=conv.pp=$pp timeout 300 python tools/vae_bench.py wan 3 > gpurun_out/r03vae/wan_b_pp$pp.log 2>&1; tail -1 gpurun_out/r03vae/wan_b_pp$pp.log; done
TUNE=conv.pp=1 timeout 200 python tools/vae_bench.py flux 3 2>&1 | tail -1
TUNE=conv.pp=0 timeout 200 python tools/vae_bench.py flux 3 2>&1 | tail -1
TUNE=conv.pp=1 timeout 300 python tools/vae_bench.py hunyuan 2 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q -m gpu 2>&1 | tail -3
