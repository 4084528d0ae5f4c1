#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats run (csv) into a short per-kernel table.

usage: python tools/prof_summary.py gpurun_out/prof1 [steps] > profiles/<name>.txt
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
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
