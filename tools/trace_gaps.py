#!/usr/bin/env python
"""Idle time BETWEEN the kernels of a denoise step, from a rocprofv3 --kernel-trace CSV: the launches of the last complete step
(delimited by the first GEMM after each scheduler step), sorted by start; gap = next start - previous end (same queue).
usage: trace_gaps.py <dir with *kernel_trace.csv> [launches per step hint]"""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mine = [i for i, r in enumerate(rows) if "gemm_bf16" in r["Kernel_Name"] or "attn_fwd" in r["Kernel_Name"] or "ln_modulate" in r["Kernel_Name"]]
# the timed steps are the tail of the run: take the last 40 % of the library's launches
lo = mine[int(len(mine) * 0.6)]
seg = rows[lo:mine[-1] + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = collections.Counter()
cnt = collections.Counter()
allg = []
for a, b in zip(seg, seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    key = (a["Kernel_Name"].split("<")[0][-28:], b["Kernel_Name"].split("<")[0][-28:])
    gaps[key] += g
    cnt[key] += 1
    allg.append(g)
allg.sort()
print(json.dumps({"launches": len(seg), "wall_ms": wall / 1e6, "sum_kernel_ms": busy / 1e6, "sum_gaps_ms": sum(allg) / 1e6,
                  "gap_share": sum(allg) / wall, "median_gap_us": allg[len(allg) // 2] / 1e3, "p90_gap_us": allg[int(len(allg) * 0.9)] / 1e3,
                  "negative_gaps(overlap)": sum(1 for g in allg if g < 0)}))
for k, v in gaps.most_common(12):
    print(f"{v / 1e6:8.3f} ms  n={cnt[k]:4d}  mean {v / cnt[k] / 1e3:6.2f} us   {k[0]} -> {k[1]}")
