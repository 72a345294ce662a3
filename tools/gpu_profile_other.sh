#!/bin/bash
# rocprofv3 kernel-trace summaries of the Qwen and Wan steps (the Flux one comes from tools/gpu_profile.sh).
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in qwen wan; do
  OUT=$R/gpurun_out/prof_$w
  rm -rf $OUT; mkdir -p $OUT
  steps=2; warm=1; [ $w = wan ] && steps=1 && warm=0
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $w -- python $R/bench.py --workload $w --steps $steps --warmup $warm --no-cpu-baseline --no-roofline --no-clip > $OUT/trace.log 2>&1; echo "$w trace $?"
  (cd $R && python tools/prof_reduce.py $OUT | head -12 | cut -c1-160)
  find $OUT -name "*kernel_trace.csv" -delete
done
