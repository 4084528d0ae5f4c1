timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "legendre or dhconv" 2>&1 | tail -5
timeout -k 10 200 python tools/kbench.py --only legendre_fwd,legendre_inv,dhconv_fwd,dhconv_dgrad,dhconv_wgrad --iters 20 2>&1 | grep x3
