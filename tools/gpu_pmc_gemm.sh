#!/bin/bash
# HBM-side traffic + MFMA-pipe occupancy of the dominant kernel of the Flux step (gemm_bf16_kernel) from separate
# rocprofv3 --pmc passes (kernel-trace only beside the counters) over the step (full depth for flux; LAYERS=3,6 = 3 double + 6 single
# blocks: the per-launch figures do not depend on the depth, and counter collection costs ~50 ms per dispatch).
# WORKLOAD=qwen: the same over a 3-block QwenImage-Edit step.
# Writes gpurun_out/pmc_gemm/r03_pmc_gemm[_qwen].json (copy to profiles/): per-launch means over the step's GEMM launches,
# FETCH_SIZE doubled per MI355X_MICROARCH.md, and the sha256 of csrc/gemm.hip the binary was built from.
set -u
R=$GRAFT_REPO_ROOT
W=${WORKLOAD:-flux}                 # flux | qwen
OUT=$R/gpurun_out/pmc_gemm_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=$([ "$W" = flux ] && echo ${ROUND:-r05}_pmc_gemm || echo ${ROUND:-r05}_pmc_gemm_$W)
export TAG W
# LAYERS: empty = the FULL-depth step (flux default since round 6: ~1300 dispatches per pass, about a minute each); "3,6" = the
# depth-reduced run of the same launches (qwen default: 60 blocks would be ~3000 dispatches per pass)
LAYERS=${LAYERS-$([ "$W" = flux ] && echo "" || echo "3,6")}
export LAYERS
CMD="python $R/bench.py --workload $W ${LAYERS:+--layers $LAYERS} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-wan"
T=${PROF_TIMEOUT:-420}
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o g -- $CMD > $OUT/fetch.log 2>&1; echo "fetch $?"
timeout $T rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/write -o g -- $CMD > $OUT/write.log 2>&1; echo "write $?"
timeout $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o g -- $CMD > $OUT/sq.log 2>&1; echo "sq $?"
cd $R
python - <<'PY'
import csv, glob, hashlib, json, os
root = os.environ["GRAFT_REPO_ROOT"]
out = root + "/gpurun_out/pmc_gemm_" + os.environ["W"] + "/"
def big(name):          # the large-tile GEMM launches of the step: the shipped 256 x 256 ping-pong kernel and (round 5) the 384 x 256 one
    return ("gemm_bf16_kernel" in name and "Cfg<256, 256, 2, 4, 5>" in name) or "gemm_bf16_x384_kernel" in name


def means(sub, pick):
    vals, durs = {}, []
    for f in glob.glob(out + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if big(r["Kernel_Name"]):
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(out + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if big(r["Kernel_Name"]):
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, (len(durs), sum(durs) / max(len(durs), 1))
f, (n, ns_f) = means("fetch", "gemm_bf16_kernel")
w, _ = means("write", "gemm_bf16_kernel")
s, _ = means("sq", "gemm_bf16_kernel")
res = {"kernel": "gemm_bf16_kernel<Cfg<256,256,2,4,5>> (ping-pong, v_mfma_f32_16x16x32_bf16) + gemm_bf16_x384_kernel (384 x 256 tiles), all epilogues",
       "command": "python bench.py --workload %s %s--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-wan" % (
           os.environ["W"], ("--layers %s " % os.environ["LAYERS"]) if os.environ.get("LAYERS") else ""),
       "source_sha256": hashlib.sha256(open(root + "/apex-studio_amd/csrc/gemm.hip", "rb").read()).hexdigest(),
       "dispatches": n, "avg_duration_ns_under_pmc": ns_f}
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    res.update(FETCH_SIZE_KB_mean=f["FETCH_SIZE"], WRITE_SIZE_KB_mean=w["WRITE_SIZE"],
               fetch_correction="gfx950 rocprofv3 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): reads = 2 x FETCH_SIZE",
               traffic_bytes_per_launch=int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024))
if "SQ_VALU_MFMA_BUSY_CYCLES" in s:
    res["SQ_VALU_MFMA_BUSY_CYCLES_mean"] = s["SQ_VALU_MFMA_BUSY_CYCLES"]
    res["SQ_BUSY_CYCLES_mean"] = s.get("SQ_BUSY_CYCLES")
    if s.get("SQ_BUSY_CYCLES"):
        res["mfma_busy_over_sq_busy"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_BUSY_CYCLES"]
    if "GRBM_GUI_ACTIVE" in w:
        res["GRBM_GUI_ACTIVE_mean"] = w["GRBM_GUI_ACTIVE"]
        res["mfma_pipe_busy_fraction"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (w["GRBM_GUI_ACTIVE"] / 8 * 1024)
        res["mfma_pipe_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs): share of SIMD-cycles with the matrix pipe busy"
    if s.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_share"] = s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open(out + os.environ["TAG"] + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
