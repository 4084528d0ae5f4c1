#!/usr/bin/env python
"""Needs a build with MK_EXTRA_HIPCC_FLAGS=-DMK_PCE_STAMPS.  In-kernel timeline of the pixel-column engine (MK_PCE_DBG=1): s_memtime stamps of workgroup 0 on its fourth tile."""
import ctypes
import os
import sys

os.environ["MK_PCE_DBG"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import _lib, ops  # noqa: E402

M, K, P = 384, 384, 721 * 1440
dev = torch.device("cuda:0")
w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
x = torch.randn(1, K, P, device=dev).bfloat16()
img = ops.pce_pack(w)
for _ in range(3):
    y = ops.pce_gemm(x, img, M)
torch.cuda.synchronize()
buf = np.zeros(512, dtype=np.uint64)
_lib.check(_lib.load().mk_pce_debug_stamps(buf.ctypes.data), "stamps")
st = buf.reshape(8, 64).astype(np.int64)
t0 = st[:, 0].min()
names = ["tile"]
for ph in range(2):
    names += [f"p{ph}bar", f"p{ph}frags"] + [f"p{ph}g{g}{k}" for g in range(6) for k in ("", "w", "b")]
names += ["loopend", "drain", "epi_end"]
w = max(len(n) for n in names)
print("stamp".ljust(w) + "  " + " ".join(f"w{i:<6d}" for i in range(8)) + "  (ticks since the first stamp; delta of wave 0 in brackets)")
prev = None
for i in range(64):
    if i >= len(names):
        names.append(f"#{i}")
    if st[:, i].max() == 0:
        break
    d = "" if prev is None else f"[{int(st[0, i] - prev)}]"
    prev = st[0, i]
    print(f"{names[i]:{w}s}  " + " ".join(f"{int(v - t0):<7d}" for v in st[:, i]) + "  " + d)
