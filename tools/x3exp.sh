timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv1x1_fwd" 2>&1 | tail -15
timeout -k 10 200 python tools/kbench.py --only conv_fwd --iters 10 2>&1 | grep conv1x1
