#!/bin/bash
# bash tools/attn_w64_variants.sh "<generator options of variant 1>" ["<variant 2>" ... up to 6]
# REAL variants of the generated w64 attention loop (correct results, no trace) in a side library next to the shipped loop:
# attn.w64 = 1 is the shipped loop, 2.. = the variants in order.  Timed and checked by tools/attn_w64_variants.py:
#   APEX_MI355_LIB=tools/ubench/bin/libapex_w64var.so python tools/attn_w64_variants.py
# W64VAR_FLAGS=-DAPEXMI_ATTN_W64_VAR_DMA_WAVE=1 builds the variants' shell for --opt=dma=wave loops (ALL variants must carry that option).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
D=$ROOT/apex-studio_amd/csrc/w64_ablate
mkdir -p $BIN $D
: > $BIN/w64var_variants.txt
for i in 1 2 3 4 5 6; do
    opts=${!i:-}
    python $ROOT/tools/gen_attn_w64.py --out=$D/v$i.inc $opts > /dev/null || exit 1
    echo "$((i + 1)): ${opts:-shipped options}" >> $BIN/w64var_variants.txt
done
cat $BIN/w64var_variants.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_ATTN_W64_ABLATE=1 ${W64VAR_FLAGS:-} -c $ROOT/apex-studio_amd/csrc/attention.hip -o $BIN/attn_w64var.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_w64var.so $ROOT/apex-studio_amd/csrc/runtime.o $ROOT/apex-studio_amd/csrc/gemm.o \
  $BIN/attn_w64var.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/attn_w64var.o && ls -la $BIN/libapex_w64var.so
