#!/usr/bin/env python
"""Align the GEMM dispatches of two rocprofv3 kernel traces (same model, different tiling) and print the
per-position mean duration: tools/trace_ab.py A_kernel_trace.csv B_kernel_trace.csv [launches_per_step]"""
import csv
import sys
import collections


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if "gemm_bf16_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Kernel_Name"].split("Cfg")[1][:24]))
    rows.sort()
    return rows


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 155
    n = min(len(a), len(b)) // per * per
    a, b = a[-n:], b[-n:]
    agg = collections.OrderedDict()
    for i in range(n):
        key = (i % per, a[i][2])
        agg.setdefault(key, [[], [], a[i][3], b[i][3]])
        agg[key][0].append(a[i][1])
        agg[key][1].append(b[i][1])
    # collapse identical (tiles) positions of the periodic block structure
    coll = collections.OrderedDict()
    for (pos, tiles), (da, db, na, nb) in agg.items():
        k = (tiles, na, nb)
        coll.setdefault(k, [0, 0.0, 0.0])
        coll[k][0] += 1
        coll[k][1] += sum(da) / len(da)
        coll[k][2] += sum(db) / len(db)
    print(f"{'tiles':>6} {'count':>5} {'A us':>9} {'B us':>9} {'B/A':>6}   A={a[0][3]}")
    ta = tb = 0
    for (tiles, na, nb), (c, sa, sb) in coll.items():
        print(f"{tiles:6d} {c:5d} {sa / c / 1e3:9.1f} {sb / c / 1e3:9.1f} {sb / sa:6.3f}   {na} | {nb}")
        ta += sa
        tb += sb
    print(f"total per step: A {ta / 1e6:.2f} ms  B {tb / 1e6:.2f} ms")
    if len(sys.argv) > 4:  # per-position listing of the first N launches of a step
        for (pos, tiles), (da, db, na, nb) in list(agg.items())[: int(sys.argv[4])]:
            print(f"pos {pos:3d} tiles {tiles:5d}  A {sum(da) / len(da) / 1e3:8.1f}  B {sum(db) / len(db) / 1e3:8.1f}  {sum(db) / sum(da):.3f}")


if __name__ == "__main__":
    main()
