#!/bin/bash
# wave-state counters + effective clock for the attention kernels (H40 S8192): attn.c4 = 0 and 1
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcattn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/attn_probe.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import apex_studio_amd
from apex_studio_amd import lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
H, S = 40, 8192
q = torch.randn(1, H, S, 128, generator=g, device="cuda").to(torch.bfloat16)
k = torch.randn(1, H, S, 128, generator=g, device="cuda").to(torch.bfloat16)
vt = torch.randn(1, H, 128, S, generator=g, device="cuda").to(torch.bfloat16)
o = torch.empty(1, S, H, 128, device="cuda", dtype=torch.bfloat16)
for c4 in (0, 1):
    lib.tune_set("attn.c4", c4)
    for _ in range(3):
        ops.attention_prepared(q, k, vt, o, S)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python /tmp/attn_probe.py > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python /tmp/attn_probe.py > $OUT/p2.log 2>&1; echo "p2 $?"
python - <<'PY'
import csv, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcattn/"
for p in ("p1", "p2"):
    dur = {}
    for r in csv.DictReader(open(root + f"{p}/{p}_kernel_trace.csv")):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(root + f"{p}/{p}_counter_collection.csv")):
        if "attn" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg.setdefault(k, collections.defaultdict(list))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["ns"].append(dur[r["Dispatch_Id"]])
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print(k, " ".join(f"{n}={v:.4g}" for n, v in sorted(m.items())))
        if "SQ_WAVE_CYCLES" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            print(f"   clk={cyc / m['ns']:.3f}GHz mfma_busy/simd={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} wait_any={m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst={m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} active={m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
PY
