for i in 1 2; do
echo "== eager"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-170
echo "== graph"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --graph 2>/dev/null | tail -1 | cut -c1-170
done
echo "== eager+timing"; timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-170
nproc; cat /proc/loadavg
