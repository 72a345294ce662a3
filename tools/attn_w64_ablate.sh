#!/bin/bash
# bash tools/attn_w64_ablate.sh build : timing-only variants of the generated w64 attention loop in a side library, every one with
# the per-phase cycle accumulators (--trace); attn.w64 = 2 is the full loop, 3.. = VARIANTS below (WRONG results by design)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BIN=$ROOT/tools/ubench/bin
D=$ROOT/apex-studio_amd/csrc/w64_ablate
mkdir -p $BIN $D
V2=${V2:---no-fill_x --no-fill_y}
V3=${V3:---no-exp}
V4=${V4:---no-add}
V5=${V5:---no-fma}
V6=${V6:---no-dma}
python $ROOT/tools/gen_attn_w64.py --out=$D/v1.inc --trace
python $ROOT/tools/gen_attn_w64.py --out=$D/v2.inc --trace $V2
python $ROOT/tools/gen_attn_w64.py --out=$D/v3.inc --trace $V3
python $ROOT/tools/gen_attn_w64.py --out=$D/v4.inc --trace $V4
python $ROOT/tools/gen_attn_w64.py --out=$D/v5.inc --trace $V5
python $ROOT/tools/gen_attn_w64.py --out=$D/v6.inc --trace $V6
echo "2: full | 3: $V2 | 4: $V3 | 5: $V4 | 6: $V5 | 7: $V6" > $BIN/w64abl_variants.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAPEXMI_ATTN_W64_ABLATE=1 -DAPEXMI_ATTN_TRACE=1 -c $ROOT/apex-studio_amd/csrc/attention.hip -o $BIN/attn_w64abl.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $BIN/libapex_w64abl.so $ROOT/apex-studio_amd/csrc/runtime.o $ROOT/apex-studio_amd/csrc/gemm.o \
  $BIN/attn_w64abl.o $ROOT/apex-studio_amd/csrc/elementwise.o $ROOT/apex-studio_amd/csrc/conv.o && rm $BIN/attn_w64abl.o && ls -la $BIN/libapex_w64abl.so
