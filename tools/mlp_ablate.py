#!/usr/bin/env python
"""Time the fused node's forward launch (384 -> 768 -> 384 at 721 x 1440) with the library given by MK_LIB_OVERRIDE (ablation builds:
tools/build_variant.sh ablN pce_mlp.hip -DMK_MLP_ABL=N)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
K1, Hd, M, P = 384, 768, 384, 721 * 1440
x = torch.randn(1, K1, P, device=dev).to(torch.bfloat16)
w1 = torch.randn(Hd, K1, device=dev) * (2.0 / K1) ** 0.5
w2 = torch.randn(M, Hd, device=dev) * (1.0 / Hd) ** 0.5
b1 = torch.randn(Hd, device=dev)
pf = ops.pce_mlp_pack(w1, False, w2, False)
for _ in range(2):
    ops.pce_mlp(x, pf, 0, b1=b1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(8):
    ops.pce_mlp(x, pf, 0, b1=b1)
e.record()
torch.cuda.synchronize()
print(f"{os.environ.get('MK_LIB_OVERRIDE', 'default'):40s} fwd {s.elapsed_time(e) / 8:7.3f} ms")
