#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) as a per-kernel stats CSV.

  python tools/rocpd_stats.py gpurun_out/prof/flux_results.db profiles/r01_flux_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{100.0 * r[2] / total:.2f}"])
    for r in rows[:12]:
        print(f"{100.0 * r[2] / total:6.2f}%  calls={r[1]:6d}  avg={r[3] / 1e3:9.1f} us  {r[0][:90]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
