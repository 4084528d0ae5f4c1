#!/usr/bin/env python
"""Needs a build with MK_EXTRA_HIPCC_FLAGS=-DMK_MLP_STAMPS.  In-kernel timeline of the fused MLP node (MK_MLP_DBG=1): s_memtime
stamps of workgroup 0 on its second tile: per iteration A entry, B waits done, C barrier passed, D issues done."""
import os
import sys

os.environ["MK_MLP_DBG"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import _lib, ops  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
K1, Hd, M, P = 384, 768, 384, 721 * 1440
dev = torch.device("cuda:0")
w1 = torch.randn(Hd, K1, device=dev) / K1 ** 0.5
w2 = torch.randn(M, Hd, device=dev) / Hd ** 0.5
x = torch.randn(1, K1, P, device=dev).bfloat16()
if mode == 0:
    packed = ops.pce_mlp_pack(w1, False, w2, False)
    run = lambda: ops.pce_mlp(x, packed, 0)          # noqa: E731
else:
    packed = ops.pce_mlp_pack(w2, True, w1, True)
    pre = torch.randn(1, Hd, P, device=dev).bfloat16()
    run = lambda: ops.pce_mlp(x, packed, 1, pre=pre)   # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(512, dtype=np.uint64)
_lib.check(_lib.load().mk_pce_mlp_debug_stamps(buf.ctypes.data), "stamps")
st = buf.reshape(8, 64).astype(np.int64)
t0 = st[:, 0].min()
names = ["tile"]
for it in range(31):
    names += [f"it{it}A", f"it{it}B", f"it{it}C", f"it{it}D"]
print("stamp     " + " ".join(f"w{i:<7d}" for i in range(8)) + "  (ticks since the first stamp; delta of wave 0 / wave 3 in brackets)")
prev = None
for i in range(64):
    if st[:, i].max() == 0:
        break
    d = "" if prev is None else f"[{int(st[0, i] - prev[0])} / {int(st[4, i] - prev[4])}]"
    prev = st[:, i].copy()
    print(f"{names[i] if i < len(names) else '#' + str(i):9s} " + " ".join(f"{int(v - t0):<8d}" for v in st[:, i]) + "  " + d)
