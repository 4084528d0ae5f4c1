for mt in 2 4; do echo "== MK_WGRAD_MT=$mt"; MK_WGRAD_MT=$mt timeout -k 10 200 python tools/kbench.py --only conv_wgrad --iters 10 2>&1 | grep conv1x1; done
MK_WGRAD_MT=4 timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -k wgrad 2>&1 | tail -2
