#!/bin/bash
# Rehearse bench.py's N = 4 path on a one-GPU box: four ranks on cuda:0, gloo transport (everything but the RCCL wire).
export MK_BENCH_BACKEND=gloo MK_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python bench.py --gpus 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
