#!/usr/bin/env python
"""Per-step wall time of the first 24 training steps in a fresh process (allocator / clock settling)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from makani_amd import ops
from makani_amd.optim import FusedAdam
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SphericalFourierNeuralOperatorNet(**bench.CONFIG).to(dev)
opt = FusedAdam(net.parameters(), lr=1e-4)
inp = torch.randn(1, 73, 721, 1440, device=dev)
tar = torch.randn(1, 73, 721, 1440, device=dev)
w = torch.ones(721, device=dev)
import gc
if os.environ.get("MK_GC_OFF") == "1":
    gc.disable()
ts = []
for i in range(24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = net(inp)
    loss = ops.weighted_mse(pred, tar, w, 1e-6)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{t:.1f}" for t in ts))
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "num hipMalloc segments", torch.cuda.memory_stats()["segment.all.current"])
