#!/bin/bash
# Ablation of the free-running ring GEMM schedule (gemm.large = 9): what the K-loop costs without its barriers / vmcnt waits / LDS-DMA
# pieces / fragment reads.  Builds one library per mask HERE (hipcc cross-compiles), under tools/ubench/bin/ (ships with the snapshot);
#   bash tools/gemm_ring_ablate.sh build            (in the build container)
#   bash tools/gemm_ring_ablate.sh run              (on the GPU box: gpurun_out/r04_gemm_ring_ablate.log)
# Results of the ablated builds are WRONG by construction; only their timing is read.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
MASKS="1 3 7 15 4 8"
if [ "${1:-build}" = "build" ]; then
  mkdir -p $BIN
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_GEMM_ABLATE=$m -c $ROOT/apex-studio_amd/csrc/gemm.hip -o $BIN/gemm_abl$m.o &
  done
  wait
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_abl$m.so $ROOT/apex-studio_amd/csrc/runtime.o $BIN/gemm_abl$m.o \
      $ROOT/apex-studio_amd/csrc/attention.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/gemm_abl$m.o
  done
  ls -la $BIN
else
  cd $ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_gemm_ring_ablate.log
  for m in 0 $MASKS; do
    lib=""; [ $m != 0 ] && lib=$BIN/libapex_abl$m.so
    echo "== ablate mask $m (1 no barriers, 2 no vmcnt waits, 4 no LDS-DMA, 8 no fragment reads)" >> gpurun_out/r04_gemm_ring_ablate.log
    APEX_MI355_LIB=$lib SEQ_ONLY="7,9" timeout 300 python tools/gemm_seq_bench.py 2>/dev/null | tail -1 >> gpurun_out/r04_gemm_ring_ablate.log
  done
  cat gpurun_out/r04_gemm_ring_ablate.log
fi
