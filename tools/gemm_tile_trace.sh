#!/bin/bash
# bash tools/gemm_tile_trace.sh build   (build container: the -DAPEXMI_GEMM_TRACE=1 side library under tools/ubench/bin/, ships with the snapshot)
# bash tools/gemm_tile_trace.sh run     (GPU box: gpurun_out/r04_gemm_tile_trace.log)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
if [ "${1:-build}" = "build" ]; then
  mkdir -p $BIN
  for v in trace:0; do
    n=${v%%:*}; nt=${v##*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_ATTN_TRACE=1 -c $ROOT/apex-studio_amd/csrc/attention.hip -o $BIN/attn_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_CONV_TRACE=1 -c $ROOT/apex-studio_amd/csrc/conv.hip -o $BIN/conv_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_GEMM_TRACE=1 -c $ROOT/apex-studio_amd/csrc/gemm.hip -o $BIN/gemm_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_$n.so $ROOT/apex-studio_amd/csrc/runtime.o $BIN/gemm_$n.o \
      $BIN/attn_$n.o $ROOT/apex-studio_amd/csrc/elementwise.o $BIN/conv_$n.o && rm $BIN/gemm_$n.o $BIN/attn_$n.o $BIN/conv_$n.o
  done
else
  cd $ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_gemm_tile_trace.log
  for n in trace; do
    echo "== $n (trace_nt: non-temporal output stores)" >> gpurun_out/r04_gemm_tile_trace.log
    APEX_MI355_LIB=$BIN/libapex_$n.so timeout 300 python tools/gemm_tile_trace.py 2>&1 | tail -7 >> gpurun_out/r04_gemm_tile_trace.log
    APEX_MI355_LIB=$BIN/libapex_$n.so timeout 300 python tools/attn_tile_trace.py 2>&1 | tail -3 | tee gpurun_out/r04_attn_tile_trace.log
    APEX_MI355_LIB=$BIN/libapex_$n.so timeout 300 python tools/conv_tile_trace.py 2>&1 | tail -5 | tee gpurun_out/r04_conv_tile_trace.log
  done
  cat gpurun_out/r04_gemm_tile_trace.log
fi
