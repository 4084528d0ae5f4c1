#!/usr/bin/env python
"""One production weight-gradient shape, a few launches of mk_conv1x1_wgrad: the target of tools/wgrad_traffic.sh's PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import _lib, ops  # noqa: E402

O, I, P = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
lib = _lib.load()
gy = torch.randn(1, O, P, device=dev).bfloat16()
x = torch.randn(1, I, P, device=dev).bfloat16()
gw = torch.zeros(O, I, device=dev)
for _ in range(6):
    lib.mk_conv1x1_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), 1, O, I, P, ops._stream())
torch.cuda.synchronize()
