#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.
usage: python tools/pmc_summary.py gpurun_out/pmc_*  [substring filter on kernel name]"""
import csv
import glob
import os
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
filt = [a for a in sys.argv[1:] if not os.path.isdir(a)]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if filt and not any(x in name for x in filt):
                continue
            short = name.split("(")[0].replace("void (anonymous namespace)::", "")[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
