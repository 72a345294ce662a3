#!/usr/bin/env python
"""Per-kernel means of rocprofv3 counter_collection CSVs (one row per dispatch x counter)."""
import csv
import collections
import glob
import sys


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = name.split("(")[0][-60:] + "|grid=" + r["Grid_Size"]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        if "gemm" not in k and "attn" not in k:
            continue
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        print("==", k)
        print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    print(f"   {n}/WAVE_CYCLES = {c[n] / wc:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
            print(f"   MFMA_BUSY/BUSY_CYCLES = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print(f"   LDS conflict share = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
