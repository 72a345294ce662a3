#!/bin/bash
# unrolled serial slab kernel: VAE tests, then same-box A/B of the decodes (previous conv object vs this tree, alternating processes)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_vae.py tests/test_gpu_taehv.py tests/test_gpu_stage_parity.py tests/test_gpu_like_for_like.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04_conv_unroll_tests.log
: > gpurun_out/r04_ab_conv_unroll.log
for which in wan hunyuan flux; do
  for i in 1 2; do
    for l in prev new; do
      lib=""; [ $l = prev ] && lib=$PWD/tools/ubench/bin/libapex_prev.so
      echo "== $which $l (round $i)" >> gpurun_out/r04_ab_conv_unroll.log
      APEX_MI355_LIB=$lib timeout 600 python tools/vae_bench.py $which 3 2>&1 | tail -1 >> gpurun_out/r04_ab_conv_unroll.log
    done
  done
done
cat gpurun_out/r04_ab_conv_unroll.log
