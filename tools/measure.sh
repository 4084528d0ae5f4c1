#!/bin/bash
# Round measurement set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats, PMC traffic.
# usage: bash tools/measure.sh <tag>      -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.txt, <tag>_traffic.json, <tag>_pmc_util.json
set -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out
timeout -k 10 500 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
tail -1 $out/${tag}_bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o runc -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/${tag}_prof.log 2>&1 || exit 2
echo "kernel trace done"
python3 tools/prof_summary.py $out/${tag}_prof 5 > $out/${tag}_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/${tag}_traffic/pmc_$c -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/${tag}_pmc_$c.log 2>&1 || exit 3
  echo "pmc $c done"
done
python3 tools/traffic_summary.py $out/${tag}_traffic > $out/${tag}_traffic.json
echo "traffic summary done"
for c in MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $out/${tag}_traffic/pmc_$c -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/${tag}_pmc_$c.log 2>&1 || exit 4
  echo "pmc $c done"
done
python3 tools/util_summary.py $out/${tag}_traffic > $out/${tag}_pmc_util.json
echo "utilisation summary done"
