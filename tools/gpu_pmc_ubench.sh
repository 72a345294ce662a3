#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / duration) and MFMA busy share of a microbenchmark's kernels
set -u
BIN=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcub
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- $GRAFT_REPO_ROOT/$BIN > $OUT/p1.log 2>&1; echo "p1 $?"
python - <<'PY'
import csv, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcub/p1/"
dur = {}
for r in csv.DictReader(open(root + "p1_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.OrderedDict()
for r in csv.DictReader(open(root + "p1_counter_collection.csv")):
    k = r["Kernel_Name"][:60]
    agg.setdefault(k, collections.defaultdict(list))
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["ns"].append(dur[r["Dispatch_Id"]])
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:62s} ms={m['ns'] / 1e6:7.3f} clk={cyc / m['ns']:.3f}GHz mfma_busy/simd={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
PY
