#!/bin/bash
# Where the wave-cycles of the direct-convolution kernel go: rocprofv3 --pmc passes (kernel-trace only beside the counters) over
# one layer (CONV_CASE, default the 96->96 3x3x3 full-resolution layer of the Wan decoder) for both schedules (conv.pp 0 / 1).
# Writes gpurun_out/pmc_conv/summary.json: per-launch means of every counter, per schedule.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_conv
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=${PROF_TIMEOUT:-240}
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
      "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
      "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_BF16 SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAVES")
for pp in 0 1; do
  i=0
  for s in "${SETS[@]}"; do
    CONV_PP=$pp timeout $T rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/pp${pp}_$i -o g -- python $R/tools/conv_one.py > $OUT/pp${pp}_$i.log 2>&1
    echo "pp=$pp set $i rc=$?"
    i=$((i+1))
  done
done
cd $R
python - <<'PY'
import csv, glob, json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_conv/"
res = {}
for pp in (0, 1):
    vals = {}
    for f in glob.glob(out + f"pp{pp}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3d_slab" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    res[f"pp{pp}"] = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
json.dump(res, open(out + "summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
