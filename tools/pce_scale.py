#!/usr/bin/env python
"""Engine launch time against the number of pixel tiles per workgroup: separates the fixed cost of a launch (weight
preload, ring prologue, drain) from the per-tile cost.  usage: pce_scale.py M K [addend]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops  # noqa: E402


def main():
    M, K = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda:0")
    w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    img = ops.pce_pack(w)
    for tiles in (1, 2, 4, 8, 16, 32, 64):
        P = 64 * 256 * tiles
        x = torch.randn(1, K, P, device=dev).bfloat16()
        add = torch.randn(1, M, P, device=dev).bfloat16() if "addend" in sys.argv else None
        fn = lambda: ops.pce_gemm(x, img, M, addend=add)
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 8)
        t = sorted(ts)[2]
        print(f"{tiles:3d} x 64 px per CU: {t * 1e3:8.1f} us  ({t * 1e3 / tiles:6.2f} us per tile, {2 * (M + K) * P / t / 1e6:6.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
