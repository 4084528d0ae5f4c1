timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
timeout -k 10 200 python tools/kbench.py --only conv_wgrad --iters 10 2>&1 | grep conv1x1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-200
