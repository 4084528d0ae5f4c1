#!/usr/bin/env python
"""Fused conv -> GELU -> conv node (mk_pce_mlp) against the pair of engine launches it replaces, production shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makani_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=8):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (K1, Hd, M, P) in ((384, 768, 384, 721 * 1440), (384, 768, 384, 240 * 480), (73, 384, 384, 721 * 1440), (384, 384, 73, 721 * 1440)):
    x = torch.randn(1, K1, P, device=dev).to(torch.bfloat16)
    gy = torch.randn(1, M, P, device=dev).to(torch.bfloat16)
    w1 = torch.randn(Hd, K1, device=dev) * (2.0 / K1) ** 0.5
    w2 = torch.randn(M, Hd, device=dev) * (1.0 / Hd) ** 0.5
    b1 = torch.randn(Hd, device=dev)
    pf, pb = ops.pce_mlp_pack(w1, False, w2, False), ops.pce_mlp_pack(w2, True, w1, True)
    i1, i2 = ops.pce_pack(w1), ops.pce_pack(w2)
    i2t, i1t = ops.pce_pack(w2, transpose=True), ops.pce_pack(w1, transpose=True)
    y, pre = ops.pce_mlp(x, pf, 0, b1=b1)
    t_f = timeit(lambda: ops.pce_mlp(x, pf, 0, b1=b1, want_row_sums=True))
    t_b = timeit(lambda: ops.pce_mlp(gy, pb, 1, pre=pre, want_mid_sums=True))

    def unfused_f():
        h, p_ = ops.pce_gemm(x, i1, Hd, bias=b1, want_pre=True, gelu=True)
        return ops.pce_gemm(h, i2, M, want_row_sums=True)

    def unfused_b():
        gp, _ = ops.pce_gemm(gy, i2t, Hd, aux_in=pre, want_row_sums=True)
        return ops.pce_gemm(gp, i1t, K1)
    t_uf, t_ub = timeit(unfused_f), timeit(unfused_b)
    h = torch.nn.functional.gelu(pre.float()).to(torch.bfloat16)
    t_w = timeit(lambda: ops.conv1x1_wgrad_raw(gy, h))
    t_wa = timeit(lambda: ops.conv1x1_wgrad_raw(gy, pre, x_gelu=True))
    gb = lambda rows: 2.0 * rows * P / 1e9  # noqa: E731
    print(f"{K1:4d}->{Hd:4d}->{M:4d} P={P:8d}: fwd fused {t_f:6.3f} ms ({gb(K1 + Hd + M) / t_f:5.2f} TB/s) vs pair {t_uf:6.3f} | "
          f"bwd fused {t_b:6.3f} ms ({gb(M + 2 * Hd + K1) / t_b:5.2f} TB/s) vs pair {t_ub:6.3f} | wgrad2 {t_w:6.3f} ms, activated {t_wa:6.3f}",
          flush=True)
