#!/usr/bin/env python
"""Ablation timings of one engine shape: re-runs itself with MK_PCE_EXP = each given bit set (results are wrong by design).

usage: pce_ablate.py M K [full|low] [addend] -- bits: 1 no epilogue, 2 no MFMA, 8 no X DMA, 16 no fragment reads"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(M, K, res, addend):
    import torch
    sys.path.insert(0, ROOT)
    from makani_amd import ops
    P = 721 * 1440 if res == "full" else 240 * 480
    dev = torch.device("cuda:0")
    w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    x = torch.randn(1, K, P, device=dev).bfloat16()
    add = torch.randn(1, M, P, device=dev).bfloat16() if addend else None
    img = ops.pce_pack(w)
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
    print(f"{sorted(ts)[2]:.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1")
        sys.exit(0)
    M, K = int(sys.argv[1]), int(sys.argv[2])
    res = sys.argv[3] if len(sys.argv) > 3 else "full"
    addend = "1" if "addend" in sys.argv else "0"
    for exp in (0, 1, 2, 8, 16, 1 + 2, 1 + 8, 2 + 16, 1 + 2 + 16, 1 + 2 + 8 + 16):
        env = dict(os.environ, MK_PCE_EXP=str(exp))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(M), str(K), res, addend], env=env,
                           capture_output=True, text=True, timeout=300)
        print(f"MK_PCE_EXP={exp:2d}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
