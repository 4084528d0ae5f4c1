#!/usr/bin/env python
"""How long the Python thread needs to ISSUE one training step (no waiting for the GPU) vs the GPU time of the step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from makani_amd.optim import FusedAdam
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SphericalFourierNeuralOperatorNet(**bench.CONFIG).to(dev)
opt = FusedAdam(net.parameters(), lr=1e-4)
inp = torch.randn(1, 73, 721, 1440, device=dev)
tar = torch.randn(1, 73, 721, 1440, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = net(inp)
    loss = ((pred.float() - tar) ** 2).mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
issue, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append(1e3 * (t1 - t0))
    total.append(1e3 * (t2 - t0))
print(f"CPU issue time per step: {sorted(issue)[len(issue)//2]:.1f} ms (min {min(issue):.1f}); "
      f"issue + GPU drain: {sorted(total)[len(total)//2]:.1f} ms")
