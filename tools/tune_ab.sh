#!/bin/bash
# Step-level A/B of tune settings on ONE box: bash tools/tune_ab.sh <workload> <rounds> <key=value|-> <key=value|-> ...   ("-" = defaults)
W=${1:-flux}; N=${2:-2}; shift 2
for r in $(seq 1 $N); do
  for T in "$@"; do
    if [ "$T" = "-" ]; then A=""; else A="--tune $T"; fi
    python bench.py --workload $W $A --no-cpu-baseline --no-clip --no-wan 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('$T', '$W', 'ms_per_step', round(d['ms_per_step'],2), {n:round(v['ms_per_step'],2) for n,v in k.items()})"
  done
done
