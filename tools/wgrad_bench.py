#!/usr/bin/env python
"""Microbenchmark of the two weight-gradient kernels (mk_conv1x1_wgrad: 128x128 blocks + slabs; mk_conv1x1_wgrad_os:
output-stationary) on the production shapes, against torch (hipBLASLt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import _lib, ops  # noqa: E402

FULL, LOW = 721 * 1440, 240 * 480
SHAPES = [("384x384 full", 384, 384, FULL), ("384x73 full", 384, 73, FULL), ("73x384 full", 73, 384, FULL),
          ("73x73 full", 73, 73, FULL), ("384x384 low", 384, 384, LOW), ("768x384 low", 768, 384, LOW),
          ("384x768 low", 384, 768, LOW)]


def timeit(fn, rounds=5, reps=8):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for name, O, I, P in SHAPES:
        gy = torch.randn(1, O, P, device=dev).bfloat16()
        x = torch.randn(1, I, P, device=dev).bfloat16()
        gw = torch.zeros(O, I, device=dev)
        st = ops._stream()
        variants = {"blaslt": lambda: torch.mm(gy[0], x[0].t()),
                    "blocks": lambda: lib.mk_conv1x1_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), 1, O, I, P, st),
                    "os": lambda: lib.mk_conv1x1_wgrad_os(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), 1, O, I, P, st)}
        for fn in variants.values():
            fn()
        torch.cuda.synchronize()
        byt = 2.0 * (O + I) * P
        print(f"{name:14s} " + "  ".join(f"{k}: {t:.3f} ms ({byt / t / 1e6:.0f} GB/s)" for k, t in
                                        ((k, timeit(fn)) for k, fn in variants.items())), flush=True)


if __name__ == "__main__":
    main()
