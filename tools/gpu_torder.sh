cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 400 python tools/conv_torder_bench.py > gpurun_out/r03/conv_torder_bench.log 2>&1
for v in 0 1 0 1; do TUNE=conv.torder=$v timeout 300 python tools/vae_bench.py wan 3 2>&1 | tail -1 | sed "s/^/torder=$v /"; done | tee gpurun_out/r03/wan_decode_torder.log
for v in 0 1; do TUNE=conv.torder=$v timeout 300 python tools/vae_bench.py hunyuan 2 2>&1 | tail -1 | sed "s/^/torder=$v /"; done | tee -a gpurun_out/r03/wan_decode_torder.log
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_stage_parity.py tests/test_gpu_fullsize.py tests/test_gpu_end_to_end.py tests/test_gpu_f32_storage.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_profile.sh r03_vae_wan_tile_torder python $GRAFT_REPO_ROOT/tools/vae_bench.py wan-tile 1 2>&1 | tail -1; head -5 gpurun_out/prof_r03_vae_wan_tile_torder/derived.csv | cut -c1-180
