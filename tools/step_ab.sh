#!/bin/bash
# Step-level A/B of two builds of the library on ONE box: bash tools/step_ab.sh <workload> <rounds> <lib A | -> <lib B | ->
# ("-" = the shipped library).  Prints ms per step and the kernel classes' ms per step of every run, interleaved A, B, A, B, ...
W=${1:-flux}; N=${2:-2}; A=${3:--}; B=${4:--}
for r in $(seq 1 $N); do
  for L in "$A" "$B"; do
    if [ "$L" = "-" ]; then unset APEX_MI355_LIB; else export APEX_MI355_LIB=$L; fi
    python bench.py --workload $W --no-cpu-baseline --no-clip --no-wan 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('$L', '$W', 'ms_per_step', round(d['ms_per_step'],2), {n:round(v['ms_per_step'],2) for n,v in k.items()}, 'clock', d.get('roofline',{}).get('clock_ghz'))"
  done
done
