#!/bin/bash
# Timing-only ablation of the flash kernel's softmax cluster: one side library per mask (results WRONG by construction).
#   bash tools/attn_ablate.sh build     (build container)      bash tools/attn_ablate.sh run   (GPU box -> gpurun_out/r04_attn_ablate.log)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
MASKS="1 2 3 4 7"
if [ "${1:-build}" = "build" ]; then
  mkdir -p $BIN
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_ATTN_ABLATE=$m -c $ROOT/apex-studio_amd/csrc/attention.hip -o $BIN/attn_abl$m.o &
  done
  wait
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_attn_abl$m.so $ROOT/apex-studio_amd/csrc/runtime.o $ROOT/apex-studio_amd/csrc/gemm.o \
      $BIN/attn_abl$m.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/attn_abl$m.o
  done
else
  cd $ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_attn_ablate.log
  for m in 0 $MASKS; do
    lib=""; [ $m != 0 ] && lib=$BIN/libapex_attn_abl$m.so
    echo "== mask $m (1 no exp2, 2 no row max, 4 no V^T fragment reads)" >> gpurun_out/r04_attn_ablate.log
    APEX_MI355_LIB=$lib timeout 300 python tools/attn_time.py 2>/dev/null | tail -1 >> gpurun_out/r04_attn_ablate.log
  done
  cat gpurun_out/r04_attn_ablate.log
fi
