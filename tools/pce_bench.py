#!/usr/bin/env python
"""Microbenchmark: pixel-column engine (mk_pce_gemm) vs hipBLASLt (torch.mm) on the production 1x1-conv shapes.

Interleaved rounds in one process (guide rule 24), random bf16 data, HIP events on the current stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops  # noqa: E402

FULL, LOW = 721 * 1440, 240 * 480
SHAPES = [("enc0 73->384 full", 384, 73, FULL), ("384->384 full", 384, 384, FULL), ("fc1 384->768 full", 768, 384, FULL),
          ("fc2 768->384 full", 384, 768, FULL), ("dec2 384->73 full", 73, 384, FULL), ("res 73->73 full", 73, 73, FULL),
          ("384->384 low", 384, 384, LOW), ("fc1 384->768 low", 768, 384, LOW), ("fc2 768->384 low", 384, 768, LOW)]


def timeit(fn, rounds=5, reps=8):
    """Median / min over `rounds` of the mean of `reps` back-to-back launches (amortises the ~70 us an isolated
    launch pays between the event record and the kernel start)."""
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps)
    return sorted(ts)[len(ts) // 2], min(ts)


def main():
    dev = torch.device("cuda:0")
    only = sys.argv[1:]
    for name, M, K, P in SHAPES:
        if only and not any(o in name for o in only):
            continue
        w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
        x = torch.randn(1, K, P, device=dev).bfloat16()
        img = ops.pce_pack(w)
        variants = {"blaslt": lambda: torch.mm(w, x[0]), "pce": lambda: ops.pce_gemm(x, img, M)}
        if M == 768:
            bias = torch.randn(M, device=dev)
            variants["pce+bias+gelu+pre"] = lambda: ops.pce_gemm(x, img, M, bias=bias, want_pre=True, gelu=True)
        if M == 384 and K == 384:
            add = torch.randn(1, M, P, device=dev).bfloat16()
            variants["pce+addend"] = lambda: ops.pce_gemm(x, img, M, addend=add)
            variants["blaslt addmm"] = lambda: torch.addmm(add[0], w, x[0])
        for fn in variants.values():
            fn()
        torch.cuda.synchronize()
        res = {k: timeit(fn) for k, fn in variants.items()}
        flop, byt = 2.0 * M * K * P, 2.0 * (M + K) * P
        print(f"{name:22s} " + "  ".join(f"{k}: {med:.3f} ms ({flop / med / 1e9:.0f} TF/s, {byt / med / 1e6:.0f} GB/s)"
                                         for k, (med, mn) in res.items()), flush=True)


if __name__ == "__main__":
    main()
