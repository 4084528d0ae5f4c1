#!/bin/bash
# Rehearse bench.py's N = 2 path on a one-GPU box: both ranks on cuda:0, gloo transport (everything but the RCCL wire).
# bench.py starts its own ranks (torch.distributed.run child) when WORLD_SIZE is unset.
export MK_BENCH_BACKEND=gloo MK_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline
