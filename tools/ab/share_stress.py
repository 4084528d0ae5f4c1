"""Diagnostic: P independent processes share cuda:0, each repeats ONE deterministic op and compares every result bit-for-bit with
its first.  Separates "a kernel of this package races" from "the card corrupts results of processes that share it".

    PROCS=4 ITERS=300 python tools/ab/share_stress.py mk_fft torch_fft torch_mm mk_leg torch_ew
"""
import os
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
ITERS = int(os.environ.get("ITERS", "300"))


def make(what, dev):
    g = torch.Generator(device="cpu").manual_seed(5)
    if what in ("mk_fft", "mk_fft1440"):
        from makani_amd import ops
        n, k = (480, 240) if what == "mk_fft" else (1440, 181)
        tw = ops.fft_twiddles(n).to(dev)
        x = torch.randn(384, k, n, generator=g).to(dev)
        return lambda: ops.irfft_raw(ops.rfft_raw(x, tw, 241, 1.0, 1.0, 1.0, True), tw, n, 1.0, 1.0, 1.0, torch.float32, True)
    if what == "torch_fft":
        x = torch.randn(384, 240, 480, generator=g).to(dev)
        return lambda: torch.fft.irfft(torch.fft.rfft(x, dim=-1), n=480, dim=-1)
    if what == "torch_mm":
        a = torch.randn(4096, 4096, generator=g).to(dev)
        b = torch.randn(4096, 4096, generator=g).to(dev)
        return lambda: a @ b
    if what == "torch_mm_bf16":
        a = torch.randn(8192, 8192, generator=g).to(dev).bfloat16()
        b = torch.randn(8192, 8192, generator=g).to(dev).bfloat16()
        return lambda: a @ b
    if what == "torch_ew":
        x = torch.randn(384, 240, 480, generator=g).to(dev)
        return lambda: torch.nn.functional.gelu(x * 1.5 + 0.25).cumsum(-1)
    if what == "mk_leg":
        from makani_amd import ops
        tab = ops.legendre_table("legendre-gauss", 240, 240, 241, True).to(dev)
        xf = torch.complex(torch.randn(240, 241, 384, generator=g), torch.randn(240, 241, 384, generator=g)).to(dev)

        def f():
            c = ops.legendre_fwd_raw(xf, tab, 240, 0, None, True)
            l = torch.arange(240, device=dev).view(-1, 1, 1, 1)
            m = torch.arange(241, device=dev).view(1, -1, 1, 1)
            return torch.where(l >= m, torch.view_as_real(c), torch.zeros((), device=dev))       # rows l < m are not written
        return f
    if what == "mk_leginv":
        from makani_amd import ops
        tab = ops.legendre_table("legendre-gauss", 240, 240, 241, False).to(dev)
        c = torch.complex(torch.randn(240, 241, 384, generator=g), torch.randn(240, 241, 384, generator=g)).to(dev)
        return lambda: torch.view_as_real(ops.legendre_inv_raw(c, tab, 240, 0, None, True))
    if what == "mk_dhconv":
        from makani_amd import ops
        x = torch.complex(torch.randn(240, 241, 384, generator=g), torch.randn(240, 241, 384, generator=g)).to(dev)
        w = (torch.complex(torch.randn(240, 384, 384, generator=g), torch.randn(240, 384, 384, generator=g)) * 0.05).to(dev)
        return lambda: torch.view_as_real(ops.dhconv_fwd_raw(x, w, 1))
    if what in ("mk_dhdgrad", "mk_dhwgrad", "mk_dhconv_f32"):
        from makani_amd import ops
        x = torch.complex(torch.randn(240, 241, 384, generator=g), torch.randn(240, 241, 384, generator=g)).to(dev)
        w = (torch.complex(torch.randn(240, 384, 384, generator=g), torch.randn(240, 384, 384, generator=g)) * 0.05).to(dev)
        if what == "mk_dhdgrad":
            return lambda: torch.view_as_real(ops.dhconv_dgrad_raw(x, w, 1))
        if what == "mk_dhwgrad":
            return lambda: torch.view_as_real(ops.dhconv_wgrad_raw(x, x, 1))
        return lambda: torch.view_as_real(ops.dhconv_fwd_raw(x, w, 1, 0, 0, "f32"))
    if what in ("mk_pce", "mk_pce_mlp", "mk_wgrad"):
        from makani_amd import ops
        x = torch.randn(1, 384, 240 * 480, generator=g).to(dev).bfloat16()
        w = (torch.randn(384, 384, generator=g) * 0.05).to(dev)
        if what == "mk_pce":
            pk = ops.pce_pack(w)
            return lambda: ops.pce_gemm(x, pk, 384).float()
        if what == "mk_wgrad":
            return lambda: ops.conv1x1_wgrad_raw(x, x)
    if what == "mk_norm":
        from makani_amd import ops
        x = torch.randn(1, 384, 240, 480, generator=g).to(dev)
        wt, bs = torch.ones(384, device=dev), torch.zeros(384, device=dev)
        return lambda: ops.instance_norm(x, wt, bs, 1e-6, True)
    if what == "torch_norm":
        x = torch.randn(1, 384, 240, 480, generator=g).to(dev)
        return lambda: torch.nn.functional.gelu(torch.nn.functional.instance_norm(x, eps=1e-6))
    if what == "torch_copy":
        x = torch.randn(1, 384, 240, 480, generator=g).to(dev)
        return lambda: x.clone()
    if what == "mk_gelu":
        from makani_amd import ops
        x = torch.randn(1, 384, 240, 480, generator=g).to(dev)
        bs = torch.zeros(384, device=dev)
        return lambda: ops.bias_gelu(x, bs)
    if what == "mk_conv":
        from makani_amd import ops
        w = torch.randn(384, 384, generator=g).to(dev) * 0.05
        x = torch.randn(1, 384, 240 * 480, generator=g).to(dev)
        return lambda: ops.conv1x1_x3(w, x)
    raise SystemExit(f"unknown op {what}")


def worker(rank, whats, q, bar, done):
    """``victim@aggressor``: process 0 checks ``victim`` ITERS times while the others keep running ``aggressor`` until it has finished;
    a plain name: everyone checks it.  Barriers keep the phases of the processes aligned."""
    dev = torch.device("cuda:0")
    out = []
    for si, spec in enumerate(whats):
        victim, _, aggressor = spec.partition("@")
        what = victim if (rank == 0 or not aggressor) else aggressor
        f = make(what, dev)
        ref = f().clone()
        torch.cuda.synchronize()
        bar.wait()
        bad, nbad, n = 0, [], 0
        while (n < ITERS) if (rank == 0 or not aggressor) else (done.value <= si):
            y = f()
            ne = (y != ref)
            n += 1
            if bool(ne.any()):
                bad += 1
                if len(nbad) < 5:
                    nbad.append(int(ne.sum()))
        torch.cuda.synchronize()
        if rank == 0:
            done.value = si + 1
        bar.wait()
        out.append(f"proc {rank} [{spec}] {what}: {bad} of {n} results differ from the first" + (f" (elements: {nbad})" if nbad else ""))
    q.put(out)


def two_streams(spec):
    """ONE process, two streams: the victim on one, the aggressor on the other (no host synchronisation between them)."""
    victim, _, aggressor = spec.partition("@")
    dev = torch.device("cuda:0")
    fv, fa = make(victim, dev), make(aggressor, dev)
    ref = fv().clone()
    torch.cuda.synchronize()
    sv, sa = torch.cuda.Stream(), torch.cuda.Stream()
    flags = []
    for it in range(ITERS):
        with torch.cuda.stream(sa):
            for _ in range(3):
                fa()
        with torch.cuda.stream(sv):
            flags.append((fv() != ref).any())
    torch.cuda.synchronize()
    bad = int(torch.stack(flags).sum())
    print(f"one process, two streams [{spec}] {victim}: {bad} of {ITERS} results differ from the first", flush=True)


if __name__ == "__main__":
    if os.environ.get("STREAMS", "0") == "1":
        for spec in sys.argv[1:]:
            two_streams(spec)
        sys.exit(0)
    whats = sys.argv[1:] or ["mk_fft", "torch_fft", "torch_mm"]
    procs_n = int(os.environ.get("PROCS", "4"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    bar, done = ctx.Barrier(procs_n), ctx.Value("i", 0)
    procs = [ctx.Process(target=worker, args=(r, whats, q, bar, done)) for r in range(procs_n)]
    for p in procs:
        p.start()
    for _ in procs:
        print("\n".join(q.get(timeout=1200)), flush=True)
    for p in procs:
        p.join(timeout=60)
