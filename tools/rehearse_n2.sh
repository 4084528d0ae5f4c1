#!/bin/bash
# Rehearse bench.py's N = 2 path on a one-GPU box: both ranks on cuda:0, gloo transport (everything but the RCCL wire).
export MK_BENCH_BACKEND=gloo MK_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline
