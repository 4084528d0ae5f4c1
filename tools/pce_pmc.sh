#!/bin/bash
# PMC view of the pixel-column engine on the 384 -> 384 full-resolution GEMM (run on the GPU box).
export TMPDIR=/tmp
out=gpurun_out/pce_pmc
rm -rf $out; mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "MfmaUtil" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out/$tag -o run -- python3 tools/pce_bench.py "384->384 full" > $out/$tag.log 2>&1 || echo "pmc $set failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pce_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pce_kernel" in r["Kernel_Name"] and "Lb0" in r["Kernel_Name"].replace("false", "Lb0"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
