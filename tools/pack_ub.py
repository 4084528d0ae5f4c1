#!/usr/bin/env python
"""Cost of the pack / unpack copies around the distributed-SHT all-to-alls at N = 8 (no communication: one GPU)."""
import sys, os, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd.distributed import compute_split_shapes

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def pack(x, dim0, world):
    chunks = torch.split(x, compute_split_shapes(x.shape[dim0], world), dim=dim0)
    send = torch.empty(x.numel(), dtype=x.dtype, device=x.device)
    off = 0
    for c in chunks:
        n = c.numel()
        send[off:off + n].view(c.shape).copy_(c)
        off += n
    return send


def unpack(recv, shapes, dim1):
    parts, off = [], 0
    for s in shapes:
        n = int(torch.Size(s).numel())
        parts.append(recv[off:off + n].view(s))
        off += n
    return torch.cat(parts, dim=dim1)


W = 8
cases = [("T1 full-res xf: split C, gather K", (241, 91, 8, 384), 3, 1, [(241, k, 8, 48) for k in compute_split_shapes(721, W)]),
         ("T2 c: split L, gather C", (240, 241, 8, 48), 0, 3, [(30, 241, 8, 48)] * W),
         ("T1 low-res xf", (241, 30, 8, 384), 3, 1, [(241, 30, 8, 48)] * W)]
for name, shp, d0, d1, rshapes in cases:
    x = torch.randn(*shp, dtype=torch.complex64, device=dev)
    nbytes = x.numel() * 8
    tp = timeit(lambda: pack(x, d0, W))
    rn = sum(int(torch.Size(s).numel()) for s in rshapes)
    recv = torch.randn(rn, dtype=torch.complex64, device=dev)
    tu = timeit(lambda: unpack(recv, rshapes, d1))
    print(f"{name:40s} {nbytes / 1e6:7.0f} MB  pack {tp:6.3f} ms ({2 * nbytes / tp / 1e9:6.2f} TB/s r+w)  "
          f"unpack {tu:6.3f} ms ({2 * rn * 8 / tu / 1e9:6.2f} TB/s r+w)")
