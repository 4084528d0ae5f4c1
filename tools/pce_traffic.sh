#!/bin/bash
# HBM bytes (PMC FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; KiB) per engine launch on the shapes given (default: the wide
# full-resolution layer and the square one), to compare with the algorithmic 2 (K + M) P.  Run on the GPU box.
export TMPDIR=/tmp
out=gpurun_out/pce_traffic
rm -rf $out; mkdir -p $out
shapes=("${@:-fc1 384->768 full}")
[ $# -eq 0 ] && shapes=("fc1 384->768 full" "384->384 full" "fc2 768->384 full")
for sh in "${shapes[@]}"; do
  tag=$(echo "$sh" | tr -c 'a-zA-Z0-9' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/$tag/$c -o run -- python3 tools/pce_bench.py "$sh" > $out/$tag.$c.log 2>&1 || echo "pmc $c failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pce_traffic/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pce_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rd = 2 * 1024 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
        wr = 1024 * sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        print(f"{os.path.basename(d[:-1]):28s} {k:60s} launches {len(v['FETCH_SIZE']):3d} read {rd/1e6:8.1f} MB write {wr/1e6:8.1f} MB")
PY
