#!/bin/bash
# effective clock + wave-state counters for the GEMM variants named in GEMM_CFGS (8192^3)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcclk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py gemm > $OUT/p1.log 2>&1; echo "p1 $?"
python - <<'PY'
import csv, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcclk/p1/"
dur = {}
for r in csv.DictReader(open(root + "p1_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(root + "p1_counter_collection.csv")):
    if "gemm" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("Cfg")[1][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["ns"].append(dur[r["Dispatch_Id"]][0])
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    ghz = m["GRBM_GUI_ACTIVE"] / m["ns"]
    print(f"{k:42s} ns={m['ns']:.0f} clk={ghz:.3f}GHz gui_cycles={m['GRBM_GUI_ACTIVE']:.4g} mfma_busy/simd={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 1024):.3f} wait_any={m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst={m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} active={m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
PY
