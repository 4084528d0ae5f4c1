#!/usr/bin/env python
"""Micro-benchmark of the hand-written kernels at the north-star shapes (one process, interleaved rounds).

usage: python tools/kbench.py [--iters 20] [--only rfft,legendre_fwd,...] [--bc 384] [--batch 1]
Prints one line per (kernel, shape): median ms, achieved GB/s or TFLOP/s, fraction of peak.
"""
import argparse
import math
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makani_amd import ops  # noqa: E402

PEAK_TF, PEAK_GB = 157.3, 8000.0


def tri_pairs(L, M):
    return sum(max(0, min(M, l + 1)) for l in range(L))


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--bc", type=int, default=384)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    dev = torch.device("cuda:0")
    L, M = 240, 241
    T = tri_pairs(L, M)
    bc = args.bc * args.batch
    rows = []

    def report(name, shape, ms, best, work, unit):
        if unit == "flop":
            ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_TF, "TFLOP/s"
        else:
            ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_GB, "GB/s"
        rows.append((name, shape, ms, best, ach, u, ach / peak))
        print(f"{name:22s} {shape:22s} {ms:8.3f} ms (min {best:7.3f})  {ach:9.1f} {u:8s} {100 * ach / peak:5.1f}% of peak",
              flush=True)

    for (grid, K, N) in (("equiangular", 721, 1440), ("legendre-gauss", 240, 480)):
        shape = f"{bc}x{K}x{N}"
        tw = ops.fft_twiddles(N).to(dev)
        x = torch.randn(bc, K, N, device=dev)
        xb = x.to(torch.bfloat16)
        s = 2 * math.pi / N
        xf = ops.rfft_raw(x, tw, M, s, s, s)
        fft_bytes = K * bc * (4 * N + 8 * M)
        if not only or "rfft" in only:
            ms, b = timeit(lambda: ops.rfft_raw(x, tw, M, s, s, s), args.iters)
            report("rfft", shape, ms, b, fft_bytes, "byte")
            ms, b = timeit(lambda: ops.rfft_raw(xb, tw, M, s, s, s), args.iters)
            report("rfft(bf16 in)", shape, ms, b, K * bc * (2 * N + 8 * M), "byte")
            ms, b = timeit(lambda: ops.rfft_raw(xb, tw, M, s, s, s, kmajor=True), args.iters)
            report("rfft(bf16 in, [K,M])", shape, ms, b, K * bc * (2 * N + 8 * M), "byte")
        if not only or "irfft" in only:
            ms, b = timeit(lambda: ops.irfft_raw(xf, tw, N, 1.0, 1.0, 1.0), args.iters)
            report("irfft", shape, ms, b, fft_bytes, "byte")
            ms, b = timeit(lambda: ops.irfft_raw(xf, tw, N, 1.0, 1.0, 1.0, torch.bfloat16), args.iters)
            report("irfft(bf16 out)", shape, ms, b, K * bc * (2 * N + 8 * M), "byte")
            xfk = xf.permute(1, 0, 2).contiguous()
            ms, b = timeit(lambda: ops.irfft_raw(xfk, tw, N, 1.0, 1.0, 1.0, torch.bfloat16, kmajor=True), args.iters)
            report("irfft(bf16 out, [K,M])", shape, ms, b, K * bc * (2 * N + 8 * M), "byte")
            del xfk
        tabw = ops.legendre_table(grid, K, L, M, True).to(dev)
        c = ops.legendre_fwd_raw(xf, tabw, L)
        leg_flop = 4.0 * T * K * bc
        for mode in ("f32", "bf16x3"):
            if not only or "legendre_fwd" in only:
                ms, b = timeit(lambda: ops.legendre_fwd_raw(xf, tabw, L, mode=mode), args.iters)
                report(f"legendre_fwd[{mode}]", shape, ms, b, leg_flop, "flop")
            if not only or "legendre_inv" in only:
                ms, b = timeit(lambda: ops.legendre_inv_raw(c, tabw, K, mode=mode), args.iters)
                report(f"legendre_inv[{mode}]", shape, ms, b, leg_flop, "flop")
        if not only or "legendre_fwd" in only:
            xfk = xf.permute(1, 0, 2).contiguous()
            ms, b = timeit(lambda: ops.legendre_fwd_raw(xfk, tabw, L, kmajor=True), args.iters)
            report("legendre_fwd[x3,[K,M]]", shape, ms, b, leg_flop, "flop")
            del xfk
        if not only or "legendre_inv" in only:
            ms, b = timeit(lambda: ops.legendre_inv_raw(c, tabw, K, kmajor=True), args.iters)
            report("legendre_inv[x3,[K,M]]", shape, ms, b, leg_flop, "flop")
        del x, xb, xf, tabw, c

    E, B = args.bc, args.batch
    xs = torch.randn(L, M, B * E, dtype=torch.complex64, device=dev)
    w = torch.randn(L, E, E, dtype=torch.complex64, device=dev)
    gy = torch.randn(L, M, B * E, dtype=torch.complex64, device=dev)
    flop = 8.0 * E * E * T * B
    shape = f"L{L} M{M} B{B} E{E}"
    for mode in ("f32", "bf16x3"):
        if not only or "dhconv_fwd" in only:
            ms, b = timeit(lambda: ops.dhconv_fwd_raw(xs, w, B, mode=mode), args.iters)
            report(f"dhconv_fwd[{mode}]", shape, ms, b, flop, "flop")
        if not only or "dhconv_dgrad" in only:
            ms, b = timeit(lambda: ops.dhconv_dgrad_raw(gy, w, B, mode=mode), args.iters)
            report(f"dhconv_dgrad[{mode}]", shape, ms, b, flop, "flop")
        if not only or "dhconv_wgrad" in only:
            ms, b = timeit(lambda: ops.dhconv_wgrad_raw(xs, gy, B, mode=mode), args.iters)
            report(f"dhconv_wgrad[{mode}]", shape, ms, b, flop, "flop")
    if not only or "conv_wgrad" in only:
        from makani_amd import _lib
        for (O, I, P) in ((768, 384, 115200), (384, 768, 115200), (384, 384, 115200), (768, 384, 1038240),
                          (384, 768, 1038240), (384, 384, 1038240)):
            gyb = torch.randn(1, O, P, device=dev).to(torch.bfloat16)
            xb2 = torch.randn(1, I, P, device=dev).to(torch.bfloat16)
            gw = torch.zeros(O, I, device=dev)
            fn = _lib.load().mk_conv1x1_wgrad

            def run():
                _lib.check(fn(gyb.data_ptr(), xb2.data_ptr(), gw.data_ptr(), 1, O, I, P, ops._stream()), "mk_conv1x1_wgrad")
            ms, b = timeit(run, args.iters)
            rows.append(("conv_wgrad", f"{O}x{I}x{P}", ms, b))
            tf = 2.0 * O * I * P / (ms * 1e-3) / 1e12
            gb = 2.0 * (O + I) * P / (ms * 1e-3) / 1e9
            print(f"{'conv1x1_wgrad':22s} {f'{O}x{I}x{P}':22s} {ms:8.3f} ms (min {b:7.3f})  {tf:9.1f} TFLOP/s bf16  {gb:8.1f} GB/s unique",
                  flush=True)
            del gyb, xb2, gw


if __name__ == "__main__":
    main()
