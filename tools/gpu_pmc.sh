#!/bin/bash
# PMC passes (counters only, no trace domains besides kernel-trace) for the hot kernels.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py ${1:-all} > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 600 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py ${1:-all} > $OUT/p2.log 2>&1; echo "p2 $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p3 -o p3 -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py ${1:-all} > $OUT/p3.log 2>&1; echo "p3 $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE $P3 --kernel-trace --output-format csv -d $OUT/p4 -o p4 -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py ${1:-all} > $OUT/p4.log 2>&1; echo "p4 $?"
ls -R $OUT | head -40
