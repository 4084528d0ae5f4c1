#!/usr/bin/env python
"""Which torch ops launch the small copy / fill / cast kernels of one training step (torch.profiler, one GPU)."""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from makani_amd import ops  # noqa: E402
from makani_amd.optim import FusedAdam  # noqa: E402
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet  # noqa: E402

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
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and e.count > 0:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for t, c, k, sh in rows[:int(os.environ.get('MK_TRACE_ROWS', '70'))]:
    print(f"{t / 1e3:8.3f} ms  x{c:4d}  {k[:60]:60s} {sh}")
# the small launches, by the op that issued them (kernels of at most 10 us on average)
print("\n# launches of <= 10 us average, by op")
small = [(c, t, k, sh) for t, c, k, sh in rows if c > 0 and t / c <= 10.0 and not k.startswith("void ") and "anonymous namespace" not in k]
small.sort(reverse=True)
for c, t, k, sh in small[:60]:
    print(f"x{c:4d}  {t / 1e3:7.3f} ms  {k[:60]:60s} {sh}")
print(f"# kernels launched in the step: {sum(e.count for e in prof.key_averages() if e.device_time_total > 0 and ('void ' in e.key or 'anonymous namespace' in e.key or 'Memcpy' in e.key or 'Memset' in e.key))}")

if os.environ.get("MK_TRACE_FILLS", "0") == "1":
    # who issues the zero fills: python call sites of aten::fill_ / zero_ / zeros that launch a kernel
    import collections
    import traceback
    sites = collections.Counter()
    orig = {}

    def spy(name, fn):
        def inner(*a, **k):
            fr = [f for f in traceback.extract_stack()[:-1] if "/makani_amd/" in f.filename or f.filename.endswith("bench.py")]
            sites[(name,) + tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:])] += 1
            return fn(*a, **k)
        return inner
    for name in ("zeros", "zeros_like", "full", "ones"):
        orig[name] = getattr(torch, name)
        setattr(torch, name, spy(name, orig[name]))
    for name in ("zero_", "fill_", "new_zeros"):
        orig["T." + name] = getattr(torch.Tensor, name)
        setattr(torch.Tensor, name, spy("T." + name, orig["T." + name]))
    step()
    torch.cuda.synchronize()
    print("\n# python call sites of zero fills in one step")
    for k, c in sites.most_common(40):
        print(f"x{c:4d}  {' <- '.join(k)}")
