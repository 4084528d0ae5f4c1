#!/usr/bin/env python
"""1x1-conv GEMMs on an N = 8 shard (batch 8, 1/8 of the pixels each) vs the single-sample shape (one GPU)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops

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


for (O, I, P1) in ((768, 384, 115200), (384, 768, 115200), (768, 384, 1038240), (384, 384, 1038240)):
    w = (torch.randn(O, I, device=dev) / I ** 0.5).to(torch.bfloat16)
    x1 = torch.randn(1, I, P1, device=dev).to(torch.bfloat16)
    B = 8
    P8 = (P1 // B) // 8 * 8
    x8 = torch.randn(B, I, P8, device=dev).to(torch.bfloat16)
    g1 = torch.randn(1, O, P1, device=dev).to(torch.bfloat16)
    g8 = torch.randn(B, O, P8, device=dev).to(torch.bfloat16)
    t_mm = timeit(lambda: torch.mm(w, x1[0]))
    t_bmm = timeit(lambda: torch.bmm(w.unsqueeze(0).expand(B, -1, -1), x8))
    t_hip = timeit(lambda: ops.conv1x1_fwd_raw(w, x8))
    t_w1 = timeit(lambda: ops.conv1x1_wgrad_raw(g1, x1))
    t_w8 = timeit(lambda: ops.conv1x1_wgrad_raw(g8, x8))
    print(f"{O}x{I}  P {P1}: mm(B=1) {t_mm:.3f} ms | bmm(B=8) {t_bmm:.3f} ms | hip fwd(B=8) {t_hip:.3f} ms || "
          f"wgrad B=1 {t_w1:.3f} ms, B=8 {t_w8:.3f} ms", flush=True)
