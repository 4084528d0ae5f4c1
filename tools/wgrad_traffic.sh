#!/bin/bash
# HBM bytes per weight-gradient launch (PMC FETCH_SIZE x2 gfx950 correction, WRITE_SIZE) on the production shapes against the
# algorithmic 2 (O + I) P.  Run on the GPU box.
export TMPDIR=/tmp
out=gpurun_out/wgrad_traffic
rm -rf $out; mkdir -p $out
FULL=$((721*1440)); LOW=$((240*480))
for sh in "384 384 $LOW" "768 384 $LOW" "384 768 $LOW" "384 384 $FULL" "768 384 $FULL" "384 768 $FULL" "384 73 $FULL" "73 384 $FULL"; do
  tag=$(echo "$sh" | tr ' ' 'x')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/$tag/$c -o run -- python3 tools/wgrad_traffic.py $sh > $out/$tag.$c.log 2>&1 || echo "pmc $c failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/wgrad_traffic/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    o, i, p = (int(v) for v in os.path.basename(d[:-1]).split("x"))
    for k, v in acc.items():
        rd = 2 * 1024 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
        wr = 1024 * sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        print(f"{os.path.basename(d[:-1]):18s} {k[22:70]:48s} read {rd/1e6:8.1f} MB write {wr/1e6:6.1f} MB algorithmic {2*(o+i)*p/1e6:8.1f} MB")
PY
