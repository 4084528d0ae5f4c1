nproc; cat /proc/loadavg; python -c "import torch,time; a=torch.randn(1000,1000); t=time.time(); [a@a for _ in range(20)]; print('cpu matmul s', time.time()-t)"
echo "== eager (no timing)"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-170
echo "== graph (no timing)"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --graph 2>/dev/null | tail -1 | cut -c1-170
echo "== eager (default timing)"; timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-170
echo "== graph (default timing)"; timeout -k 10 200 python bench.py --no-cpu-baseline --graph 2>/dev/null | tail -1 | cut -c1-170
