#!/bin/bash
# A/B of code-alignment builds of gemm.hip on the Flux step's GEMM sequence (tools/gemm_seq_bench.py, shipped tiling only):
#   bash tools/gemm_align_ab.sh <lib> [<lib> ...]     ("-" = the shipped library); three interleaved rounds
for r in 1 2 3; do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset APEX_MI355_LIB; else export APEX_MI355_LIB=$L; fi
    echo -n "$L: "; SEQ_ONLY=${CFG:-7} python tools/gemm_seq_bench.py | tail -1
  done
done
