#!/usr/bin/env python
"""Reduce rocprofv3 CSV output (kernel_trace + counter_collection) to small per-kernel summaries that can
be committed under profiles/: kernel_stats.csv (calls, total/avg/min/max ns, %) and pmc_summary.csv
(per-kernel mean of every counter per dispatch)."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(gemm_bf16_kernel<[^(]*?>\s*>?|attn_fwd_d128_kernel<\d+>|[A-Za-z_0-9]+_kernel(?:<[^>]*>)?)", name)
    n = m.group(1) if m else name.split("(")[0]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[-110:]


def main(root):
    traces = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if traces:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in agg.values()) or 1
        with open(os.path.join(root, "kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), f"{sum(v) / len(v):.1f}", min(v), max(v), f"{100.0 * sum(v) / tot:.2f}"])
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pm[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pm:
        counters = sorted({c for d in pm.values() for c in d})
        with open(os.path.join(root, "pmc_summary.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Dispatches"] + counters)
            for k, d in sorted(pm.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
                n = max(len(v) for v in d.values())
                w.writerow([k, n] + [f"{sum(d[c]) / len(d[c]):.6g}" if c in d else "" for c in counters])
    # ---- derived per-kernel figures: HBM-side bytes per launch and achieved GB/s (FETCH_SIZE is in KB and, on gfx950,
    # counts HALF the bytes of wide coalesced reads — doubled here as MI355X_MICROARCH.md prescribes; WRITE_SIZE in KB,
    # uncalibrated), MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) and
    # the judge's SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES; L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (requests of one XCD's L2:
    # what misses goes to the fabric — Infinity Cache or HBM — and is what FETCH_SIZE tallies)
    if traces and pm:
        with open(os.path.join(root, "derived.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalMs", "AvgUs", "FetchMBPerLaunch_x2", "WriteMBPerLaunch", "AchievedGBps",
                        "FracOf8TBps", "MfmaBusyOverAllSimds", "MfmaBusyOverSqBusy", "LdsConflictShare", "L2HitRate"])
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                d = pm.get(k, {})
                mean = lambda c: (sum(d[c]) / len(d[c])) if c in d and d[c] else None   # noqa: E731
                avg_ns = sum(v) / len(v)
                fetch = mean("FETCH_SIZE")
                write = mean("WRITE_SIZE")
                fetch_mb = fetch * 2 * 1024 / 1e6 if fetch is not None else None
                write_mb = write * 1024 / 1e6 if write is not None else None
                gbps = ((fetch_mb or 0) + (write_mb or 0)) * 1e6 / avg_ns if (fetch_mb is not None or write_mb is not None) else None
                mf, gui, sqb = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE"), mean("SQ_BUSY_CYCLES")
                lc, la = mean("SQ_LDS_BANK_CONFLICT"), mean("SQ_LDS_IDX_ACTIVE")
                th, tm = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")      # MI355X_MICROARCH.md, L2: hit rate = HIT / (HIT + MISS)
                fmt = lambda x, p=3: "" if x is None else f"{x:.{p}f}"   # noqa: E731
                w.writerow([k, len(v), f"{sum(v) / 1e6:.3f}", f"{avg_ns / 1e3:.2f}", fmt(fetch_mb), fmt(write_mb), fmt(gbps, 1),
                            fmt(gbps / 8000.0 if gbps is not None else None), fmt(mf / (gui / 8 * 1024) if mf and gui else None),
                            fmt(mf / sqb if mf and sqb else None), fmt(lc / la if lc is not None and la else None),
                            fmt(th / (th + tm) if th is not None and tm is not None and th + tm > 0 else None)])
    for name in ("kernel_stats.csv", "pmc_summary.csv", "derived.csv"):
        p = os.path.join(root, name)
        if os.path.exists(p):
            print("==", name)
            print("".join(open(p).readlines()[:14]))


if __name__ == "__main__":
    main(sys.argv[1])
