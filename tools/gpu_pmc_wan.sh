#!/bin/bash
# HBM-side traffic of the attention kernel in the Wan step: separate FETCH_SIZE / WRITE_SIZE passes (kernel-trace only),
# reduced to profiles/r01_pmc_attn_wan.json by the caller.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_wan
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload wan --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-clip"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o wan -- $CMD > $OUT/fetch.log 2>&1; echo "fetch $?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o wan -- $CMD > $OUT/write.log 2>&1; echo "write $?"
cd $R
python - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_wan/"
res = {}
for name in ("fetch", "write"):
    vals, durs = [], []
    for f in glob.glob(root + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_d128" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    for f in glob.glob(root + name + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_d128" in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res[name] = dict(n=len(vals), mean_kb=sum(vals) / max(len(vals), 1), mean_ns=sum(durs) / max(len(durs), 1))
print(json.dumps(res))
json.dump(res, open(root + "summary.json", "w"))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
