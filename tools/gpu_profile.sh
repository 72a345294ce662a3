#!/bin/bash
# rocprofv3 evidence for the bench command: kernel-trace stats + separate PMC passes (no trace domains
# besides kernel-trace in the counter passes).  Summaries are written under gpurun_out/prof_final/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-clip"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o flux -- $CMD > $OUT/trace.log 2>&1; echo "trace $?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o flux -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "fetch $?"
timeout 900 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -o flux -- $CMD > $OUT/pmc_write.log 2>&1; echo "write $?"
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o flux -- $CMD > $OUT/pmc_sq.log 2>&1; echo "sq $?"
cd $R
python tools/prof_reduce.py $OUT
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/*
