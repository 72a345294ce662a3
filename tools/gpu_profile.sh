#!/bin/bash
# rocprofv3 evidence for one command: kernel-trace stats + separate PMC passes (counter passes carry kernel-trace only).
# usage (through gpurun): bash tools/gpu_profile.sh <tag> [command ...]     default command = the default bench line
# Summaries land in gpurun_out/prof_<tag>/{kernel_stats,pmc_summary,derived}.csv; copy what is to be judged to profiles/.
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-flux}; shift || true
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-wan"; fi
echo "$CMD" > $OUT/command.txt
T=${PROF_TIMEOUT:-900}
timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1; echo "trace $?"
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o t -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "fetch $?"
timeout $T rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -o t -- $CMD > $OUT/pmc_write.log 2>&1; echo "write $?"
timeout $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o t -- $CMD > $OUT/pmc_sq.log 2>&1; echo "sq $?"
timeout $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o t -- $CMD > $OUT/pmc_tcc.log 2>&1; echo "tcc $?"
cd $R
python tools/prof_reduce.py $OUT
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls $OUT
