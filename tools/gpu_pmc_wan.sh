#!/bin/bash
# HBM-side traffic + MFMA-pipe occupancy of the attention kernels in the Wan step (dominant kernel of `bench.py --workload
# wan`): separate rocprofv3 --pmc passes (kernel-trace only beside the counters) over ONE expert forward.  Writes
# gpurun_out/pmc_wan/${ROUND:-r04}_pmc_attn_wan.json (copy to profiles/) with the sha256 of csrc/attention.hip the binary was built
# from — bench.py reports `roofline.traffic` from it only while that hash still matches.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_wan
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload wan --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-clip"
T=${PROF_TIMEOUT:-600}
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o wan -- $CMD > $OUT/fetch.log 2>&1; echo "fetch $?"
timeout $T rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/write -o wan -- $CMD > $OUT/write.log 2>&1; echo "write $?"
timeout $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o wan -- $CMD > $OUT/sq.log 2>&1; echo "sq $?"
cd $R
python - <<'PY'
import csv, glob, hashlib, json, os, sys
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root)
import bench   # kernel_source_sha256: attention.hip + its generated includes, the hash bench.py matches against
out = root + "/gpurun_out/pmc_wan/"
def means(sub):
    vals, durs = {}, []
    for f in glob.glob(out + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_d128" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(out + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_d128" in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, (len(durs), sum(durs) / max(len(durs), 1))
f, (n, ns) = means("fetch")
w, _ = means("write")
s, _ = means("sq")
S, H, D, T = 75600, 40, 128, 512
alg = ((4 * H * S * D * 2) + (2 * H * S * D * 2 + 2 * H * T * D * 2)) / 2      # mean of a self- and a cross-attention launch (q, k, v, o / q, o + 512-key k, v)
res = {"kernel": "attn_fwd_d128_w64_kernel (self-attention, S 75600) and attn_fwd_d128_kernel<4> (cross-attention, 512 keys)",
       "command": "python bench.py --workload wan --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-clip",
       "source_sha256": bench.kernel_source_sha256("attention.hip"),
       "dispatches": n, "avg_duration_ns_under_pmc": ns, "algorithmic_bytes_per_launch": int(alg),
       "note": "mean over the 80 attention launches of one expert forward (40 self-attention S = 75600, 40 cross-attention Sk = 512); "
               "every q-block round of a head streams that head's K and V^T again because 4 MiB of L2 per XCD cannot hold them; "
               "the kernel is MFMA-bound, its memory-side traffic is a few % of the HBM roofline"}
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    res.update(FETCH_SIZE_KB_mean=f["FETCH_SIZE"], WRITE_SIZE_KB_mean=w["WRITE_SIZE"],
               fetch_correction="gfx950 rocprofv3 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): reads = 2 x FETCH_SIZE",
               traffic_bytes_per_launch=int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024))
if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in w:
    res["mfma_pipe_busy_fraction"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (w["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if s.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_share"] = s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open(out + os.environ.get("ROUND", "r04") + "_pmc_attn_wan.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
