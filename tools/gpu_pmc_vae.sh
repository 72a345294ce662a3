#!/bin/bash
# HBM-side traffic and matrix-pipe occupancy of a WHOLE VAE decode (default: the Wan 4 x 7-tile 720p x 81-frame decode the
# reference always takes; VAE=flux: the 1024^2 2-D decode) from separate rocprofv3 --pmc passes (kernel-trace only beside the
# counters), summed over every kernel of ONE decode.  Writes gpurun_out/pmc_vae_<vae>/<ROUND>_pmc_conv[_flux].json (copy to
# profiles/): totals, per-kernel table, FETCH_SIZE doubled per MI355X_MICROARCH.md, sha256 of csrc/conv.hip.
set -u
R=$GRAFT_REPO_ROOT
V=${VAE:-wan}
OUT=$R/gpurun_out/pmc_vae_$V
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=$([ "$V" = wan ] && echo ${ROUND:-r06}_pmc_conv || echo ${ROUND:-r06}_pmc_conv_$V)
export TAG V
CMD="python $R/tools/vae_bench.py $V 1"
T=${PROF_TIMEOUT:-900}
timeout $T rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o g -- $CMD > $OUT/trace.log 2>&1; echo "trace $?"
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o g -- $CMD > $OUT/fetch.log 2>&1; echo "fetch $?"
timeout $T rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/write -o g -- $CMD > $OUT/write.log 2>&1; echo "write $?"
timeout $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/sq -o g -- $CMD > $OUT/sq.log 2>&1; echo "sq $?"
cd $R
python - <<'PY'
import collections, csv, glob, hashlib, json, os, re
root = os.environ["GRAFT_REPO_ROOT"]
out = root + "/gpurun_out/pmc_vae_" + os.environ["V"] + "/"


def short(name):
    m = re.search(r"([A-Za-z_0-9]+_kernel)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


def load(sub):
    """per dispatch id: kernel name, counters; the vae_bench command decodes TWICE (warm-up + 1 timed): keep the second half"""
    rows = collections.OrderedDict()
    for f in glob.glob(out + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = rows.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)
    return [rows[i] for i in ids[len(ids) // 2:]]


def durations(sub):
    ds = []
    for f in glob.glob(out + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ds.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    ds.sort()
    return ds[len(ds) // 2:]


per = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("fetch", "write", "sq"):
    for d in load(sub):
        for k, v in d.items():
            if k != "name":
                per[d["name"]][k] += v
        if sub == "fetch":
            per[d["name"]]["dispatches"] += 1
for _, n, ns in durations("trace"):
    per[n]["ns"] += ns
tot = collections.defaultdict(float)
for d in per.values():
    for k, v in d.items():
        tot[k] += v
conv = {k: v for k, v in per.items() if "conv" in k}
cfetch = sum(v.get("FETCH_SIZE", 0) for v in conv.values())
res = {"what": {"wan": "Wan-2.2 3-D VAE, tiled 4 x 7 decode of [1,16,21,90,160] -> [1,3,81,720,1280] (every kernel of ONE decode)",
                "flux": "Flux 2-D VAE decode of [1,16,128,128] -> [1,3,1024,1024] (every kernel of ONE decode)"}.get(os.environ["V"], os.environ["V"]),
       "command": "python tools/vae_bench.py %s 1 (second of its two decodes)" % os.environ["V"],
       "source_sha256": hashlib.sha256(open(root + "/apex-studio_amd/csrc/conv.hip", "rb").read()).hexdigest(),
       "dispatches": int(tot["dispatches"]), "kernel_ns_sum": tot["ns"],
       "fetch_correction": "gfx950 rocprofv3 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): reads = 2 x FETCH_SIZE",
       "traffic_bytes_per_decode": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024),
       "read_bytes_per_decode": int(2 * tot["FETCH_SIZE"] * 1024), "write_bytes_per_decode": int(tot["WRITE_SIZE"] * 1024)}
if tot.get("SQ_VALU_MFMA_BUSY_CYCLES") and tot.get("GRBM_GUI_ACTIVE"):
    res["mfma_pipe_busy_fraction_all_kernels"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (tot["GRBM_GUI_ACTIVE"] / 8 * 1024)
    cg = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in conv.values())
    if cg:
        res["mfma_pipe_busy_fraction_conv_kernels"] = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in conv.values()) / (cg / 8 * 1024)
if tot.get("TCC_HIT_sum") is not None and tot.get("TCC_HIT_sum", 0) + tot.get("TCC_MISS_sum", 0) > 0:
    res["l2_hit_rate"] = tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])
res["kernels"] = [{"name": k, "dispatches": int(v.get("dispatches", 0)), "ms": round(v.get("ns", 0) / 1e6, 3),
                   "read_MB": round(2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6, 1), "write_MB": round(v.get("WRITE_SIZE", 0) * 1024 / 1e6, 1),
                   "mfma_pipe_busy": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
                   if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE") else None}
                  for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("ns", 0))][:24]
json.dump(res, open(out + os.environ["TAG"] + ".json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for k in res["kernels"][:10]:
    print(k)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
