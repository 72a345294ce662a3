#!/bin/bash
# bash tools/gemm_qk_ablate.sh build : side libraries tools/ubench/bin/libapex_qkabl<mask>.so with parts of the q / k epilogue of the fused
# QKV GEMM left out (-DAPEXMI_QK_ABL=mask; results WRONG by construction, timing only).  bash tools/gemm_qk_ablate.sh run (GPU box):
# the single block's fused launch timed with each (tools/gemm_epilogue_probe.py, arm "qkv").
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
MASKS=${MASKS:-"0 1 2 4 8 16 32 3 63"}
if [ "${1:-build}" = "build" ]; then
  mkdir -p $BIN
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_QK_ABL=$m -c $ROOT/apex-studio_amd/csrc/gemm.hip -o $BIN/gemm_qkabl$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_qkabl$m.so $ROOT/apex-studio_amd/csrc/runtime.o $BIN/gemm_qkabl$m.o \
        $ROOT/apex-studio_amd/csrc/attention.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/gemm_qkabl$m.o ) &
  done
  wait
  ls $BIN/libapex_qkabl*.so
else
  cd $ROOT
  for m in $MASKS; do
    echo -n "mask $m: "
    APEX_MI355_LIB=$BIN/libapex_qkabl$m.so ARMS=gelu,qkv timeout 200 python tools/gemm_epilogue_probe.py 2>&1 | grep -v amdgpu | tail -1
  done
fi
