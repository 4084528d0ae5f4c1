#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats run (csv, or the default rocpd sqlite .db) into a short per-kernel table.

usage: python tools/prof_summary.py gpurun_out/prof1 [steps] > profiles/<name>.txt
"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    agg = {}
    for name, dur in cur.execute("select name, duration from kernels"):
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(dur)
    tot = sum(a[1] for a in agg.values()) or 1.0
    rows = [{"Name": n, "Calls": a[0], "TotalDurationNs": a[1], "AverageNs": a[1] / a[0], "Percentage": 100.0 * a[1] / tot}
            for n, a in agg.items()]
    return sorted(rows, key=lambda r: -r["TotalDurationNs"])


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    csvs = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
    if csvs:
        path = csvs[0]
        rows = list(csv.DictReader(open(path)))
    else:
        path = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))[0]
        rows = rows_from_db(path)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time {tot / 1e6:.2f} ms over {steps} step(s) (warm-up steps included in `steps`)")
    print(f"# {'ms/step':>9} {'calls/step':>10} {'avg_us':>9} {'pct':>6}  kernel")
    for r in rows[:60]:
        t = float(r["TotalDurationNs"])
        print(f"  {t / 1e6 / steps:9.3f} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f} "
              f"{float(r['Percentage']):6.2f}  {r['Name'][:110]}")


if __name__ == "__main__":
    main()
