#!/bin/bash
# bash tools/attn_cluster_trace.sh build   (build container: the -DAPEXMI_ATTN_TRACE=2 side library, ships with the snapshot)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
mkdir -p $BIN
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_ATTN_TRACE=2 -c $ROOT/apex-studio_amd/csrc/attention.hip -o $BIN/attn_trace2.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_trace2.so $ROOT/apex-studio_amd/csrc/runtime.o $ROOT/apex-studio_amd/csrc/gemm.o \
  $BIN/attn_trace2.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/attn_trace2.o && ls -la $BIN/libapex_trace2.so
