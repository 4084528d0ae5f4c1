#!/usr/bin/env python
"""Cost of the torch-op DistributedInstanceNorm2d vs the HIP instance norm on an N = 8 local shard (one GPU, no comm)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd.layer_norm import DistributedInstanceNorm2d
from makani_amd.layers import InstanceNorm2d

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


for shape in ((8, 384, 30, 480), (8, 384, 91, 1440)):
    x = torch.randn(*shape, device=dev).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(*shape, device=dev).to(torch.bfloat16)
    for name, mod in (("torch-op distributed norm", DistributedInstanceNorm2d(384, eps=1e-6, affine=True).to(dev)),
                      ("HIP instance norm", InstanceNorm2d(384, eps=1e-6, affine=True).to(dev))):
        def run():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = mod(x)
            y.backward(g)
            x.grad = None
        print(f"{str(shape):24s} {name:28s} fwd+bwd {timeit(run):7.3f} ms", flush=True)
