#!/usr/bin/env python
"""Single GPU, no communication: the B = 2 step on one stream vs as two micro-batches on two streams (pipeline.py)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from makani_amd import ops
from makani_amd.optim import FusedAdam
from makani_amd.pipeline import MicroBatchRunner
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SphericalFourierNeuralOperatorNet(**bench.CONFIG).to(dev)
opt = FusedAdam(net.parameters(), lr=1e-4)
B = 2
inp = torch.randn(B, 73, 721, 1440, device=dev)
tar = torch.randn(B, 73, 721, 1440, device=dev)
w = torch.ones(721, device=dev)
runner = MicroBatchRunner(2)


def loss_of(sl):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = net(inp[sl])
    return ops.weighted_mse(pred, tar[sl], w, 1e-6)


def plain():
    opt.zero_grad(set_to_none=True)
    loss_of(slice(0, B)).backward()
    opt.step()


def micro():
    opt.zero_grad(set_to_none=True)
    runner.forward(lambda j: loss_of(slice(j, j + 1))).backward()
    runner.sync()
    opt.step()


import gc
for name, fn in (("one stream, B=2", plain), ("two micro-batches", micro), ("one stream, B=2", plain), ("two micro-batches", micro)):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    gc.enable()
    print(f"{name:20s} {1e3 * (time.perf_counter() - t0) / 8:7.2f} ms/step", flush=True)
