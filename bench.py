#!/usr/bin/env python
"""Headline benchmark: SFNO forward+backward samples/s, 73 channels on 721x1440 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = zero_grad -> forward under bf16 autocast (spectral path fp32, as the reference) ->
area-weighted MSE loss -> backward -> shared-gradient reduction -> Adam step, on synthetic
N(0,1) fields of the configuration ``sfno_linear_73chq_sc3_layers8_edim384``
(/root/reference/config/sfnonet.yaml:162-187: embed_dim 384, 8 layers, scale_factor 3,
dhconv, instance norm, MLP ratio 2, big skip).  N > 1: latitude sharded over ``h`` =
N ranks (spatial model parallelism with RCCL all-to-all), global batch N (weak scaling).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` for the dominant hand-written kernel (timed live with HIP events on the launch
stream inside the timed region) and ``cpu_baseline`` (the oracle on the host cores, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# bf16x3 spectral GEMMs: six v_mfma_f32_32x32x16_bf16 products per fp32-accurate multiply-add, priced against
# the dense bf16 MFMA peak (16 x the fp32 MFMA rate = 2516.8 TFLOP/s, the guide's "~2.5 PF dense") / 6
PEAK_MFMA_BF16_TFLOPS = 16 * 157.3
PEAK_MFMA_BF16X3_TFLOPS = round(PEAK_MFMA_BF16_TFLOPS / 6, 1)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak

CONFIG = dict(spectral_transform="sht", model_grid_type="equiangular", sht_grid_type="legendre-gauss",
              filter_type="linear", operator_type="dhconv", inp_shape=(721, 1440), out_shape=(721, 1440),
              scale_factor=3, inp_chans=73, out_chans=73, embed_dim=384, num_layers=8, use_mlp=True, mlp_ratio=2,
              activation_function="gelu", encoder_layers=1, pos_embed="none", normalization_layer="instance_norm",
              hard_thresholding_fraction=1.0, big_skip=True, separable=False)


# ----------------------------------------------------------------------------------------
# live per-kernel timing: wrap the raw C-ABI launchers of makani_amd.ops with event pairs
# ----------------------------------------------------------------------------------------
class KernelTimer:
    def __init__(self):
        self.records = []   # (name, work, unit, start_event, end_event)
        self.enabled = False
        self.only = None    # when set: record only these kernels (keeps the timed region's event overhead small)
        self.gemm_mode = "f32"

    def install(self):
        from makani_amd import ops
        self.gemm_mode = ops.SPECTRAL_GEMM

        def tri_pairs(lloc, mloc, l_off, m_off):
            t = 0
            for l in range(lloc):
                t += max(0, min(mloc, l_off + l - m_off + 1))
            return t

        def wrap(name, fn, work):
            def inner(*a, **k):
                if not self.enabled or (self.only is not None and name not in self.only):
                    return fn(*a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = fn(*a, **k)
                e.record()
                w = work(out, *a, **k)
                w, unit, byt = (w + (None,))[:3]            # (work, unit[, HBM bytes of a flop-counted kernel])
                self.records.append((name, w, unit, s, e, byt))
                return out
            return inner

        # algorithmic work per launch: only l >= m counted.  The spectral GEMMs are skinny (per mode m an
        # (L - m) x K x 2BC product), so each is priced against BOTH roofs -- its flops at the bf16x3 MFMA peak and its
        # compulsory HBM bytes (Fourier rows / spectrum read once, the 6-byte pre-split table rows once, output once) at
        # 8 TB/s -- and reported against the one that binds (the larger time)
        def leg_fwd_work(out, xf, table, lmax, m_off=0, mode=None, kmajor=False):
            mloc, k, bc = (xf.shape[1], xf.shape[0], xf.shape[2]) if kmajor else xf.shape
            t = tri_pairs(lmax, mloc, 0, m_off)
            return 4.0 * t * k * bc, "flop", 8.0 * mloc * k * bc + 6.0 * t * k + 8.0 * t * bc

        def leg_inv_work(out, c, table, nlat, m_off=0, mode=None, kmajor=False):
            lmax, mloc, bc = c.shape
            t = tri_pairs(lmax, mloc, 0, m_off)
            return 4.0 * t * nlat * bc, "flop", 8.0 * t * bc + 6.0 * t * nlat + 8.0 * mloc * nlat * bc

        def dh_work(out, a, w_phys, batch, l_off=0, m_off=0):      # fwd / dgrad: second operand is w [L, I, O]
            lloc, mloc, _ = a.shape
            t, i, o = tri_pairs(lloc, mloc, l_off, m_off), w_phys.shape[1], w_phys.shape[2]
            return 8.0 * i * o * t * batch, "flop", 8.0 * t * batch * (i + o) + 8.0 * i * o * lloc

        def dh_wgrad_work(out, x, gy, batch, l_off=0, m_off=0):    # wgrad: out is gw [L, I, O]
            lloc, mloc, _ = x.shape
            t, i, o = tri_pairs(lloc, mloc, l_off, m_off), out.shape[1], out.shape[2]
            return 8.0 * i * o * t * batch, "flop", 8.0 * t * batch * (i + o) + 8.0 * i * o * lloc

        def conv_wgrad_work(res, gy, x3, **_):
            # arithmetic intensity O*I/(O+I) <= 256 flop/byte at the production shapes, below the bf16 ridge point
            # (2517 TFLOP/s / 8 TB/s = 315): HBM is the roof -- both bf16 operands read once, the fp32 gradient written
            b, o, p = gy.shape
            i = x3.shape[1]
            return float(2 * (o + i) * p * b + 4 * o * i), "byte"

        def pce_work(out, x3, wimg, m, bias=None, addend=None, aux_in=None, want_pre=False, **_):
            # every 1x1 convolution of the net sits below the bf16 ridge (intensity M*K/(M+K) <= 256 flop per byte):
            # HBM is the roof -- X read once, Y written once, plus the optional second input / second output
            b, k, p = x3.shape
            rows = k + m + (m if (addend is not None or aux_in is not None) else 0) + (m if want_pre else 0)
            return float(2 * rows * p * b), "byte"

        def rfft_work(out, x, tw, mmax, *s, **kw):
            bc, k, n = x.shape
            return float(k * bc * (x.element_size() * n + 8 * mmax)), "byte"

        def irfft_work(out, xf, tw, nlon, *s, **kw):
            kmajor = kw.get("kmajor", s[4] if len(s) > 4 else False)
            m, k, bc = (xf.shape[1], xf.shape[0], xf.shape[2]) if kmajor else xf.shape
            return float(k * bc * (out.element_size() * nlon + 8 * m)), "byte"

        def irfft_sums_work(out, xf, tw, nlon, out_dtype, kmajor=False, chans=0, cpp=0, **_):
            # the inverse FFT that also delivers the row statistics of its output (norm0's first pass): same bytes
            x = out[0]
            if cpp:
                k, m = xf.shape[1], xf.shape[2]
            else:
                m, k = (xf.shape[1], xf.shape[0]) if kmajor else (xf.shape[0], xf.shape[1])
            return float(k * x.shape[0] * (x.element_size() * nlon + 8 * m)), "byte"

        def rfft_pm_work(out, x, tw, mmax, *s, **kw):        # peer-major Fourier rows (latitude-sharded transform): same bytes
            bc, k, n = x.shape
            return float(k * bc * (x.element_size() * n + 8 * mmax)), "byte"

        def irfft_pm_work(out, xf, tw, nlon, *s, **kw):
            p, k, m, bcp = xf.shape
            return float(k * out.shape[0] * (out.element_size() * nlon + 8 * m)), "byte"

        def layout_work(out, t, *a):
            return float(2 * 8 * t.numel()), "byte"

        ops.rfft_raw = wrap("rfft", ops.rfft_raw, rfft_work)
        ops.irfft_raw = wrap("irfft", ops.irfft_raw, irfft_work)
        ops.irfft_sums_raw = wrap("irfft", ops.irfft_sums_raw, irfft_sums_work)
        ops.rfft_pm_raw = wrap("rfft", ops.rfft_pm_raw, rfft_pm_work)
        ops.irfft_pm_raw = wrap("irfft", ops.irfft_pm_raw, irfft_pm_work)
        ops.legendre_fwd_raw = wrap("legendre_fwd", ops.legendre_fwd_raw, leg_fwd_work)
        ops.legendre_inv_raw = wrap("legendre_inv", ops.legendre_inv_raw, leg_inv_work)
        ops.dhconv_fwd_raw = wrap("dhconv_fwd", ops.dhconv_fwd_raw, dh_work)
        ops.dhconv_dgrad_raw = wrap("dhconv_dgrad", ops.dhconv_dgrad_raw, dh_work)
        ops.dhconv_wgrad_raw = wrap("dhconv_wgrad", ops.dhconv_wgrad_raw, dh_wgrad_work)
        ops.conv1x1_wgrad_raw = wrap("conv1x1_wgrad", ops.conv1x1_wgrad_raw, conv_wgrad_work)
        ops.pce_gemm = wrap("pce_gemm", ops.pce_gemm, pce_work)
        ops.spec_pack_raw = wrap("spec_pack", ops.spec_pack_raw, layout_work)
        ops.spec_unpack_raw = wrap("spec_unpack", ops.spec_unpack_raw, layout_work)

    def summary(self, steps):
        agg = {}
        for name, w, unit, s, e, byt in self.records:
            d = agg.setdefault(name, {"ms": 0.0, "work": 0.0, "unit": unit, "launches": 0, "bytes": 0.0})
            d["ms"] += s.elapsed_time(e)
            d["work"] += w
            d["bytes"] += byt or 0.0
            d["launches"] += 1
        out = {}
        for name, d in agg.items():
            sec = d["ms"] * 1e-3
            if d["unit"] == "flop_bf16":      # plain bf16 GEMM (1x1-conv weight gradient)
                peak, ach, unit, bound = PEAK_MFMA_BF16_TFLOPS, d["work"] / sec / 1e12, "TFLOP/s", "mfma"
            elif d["unit"] == "flop":
                peak = PEAK_MFMA_BF16X3_TFLOPS if self.gemm_mode == "bf16x3" else PEAK_MFMA_F32_TFLOPS
                ach, unit, bound = d["work"] / sec / 1e12, "TFLOP/s", "mfma"
                if d["bytes"] / (PEAK_HBM_GBS * 1e9) > d["work"] / (peak * 1e12):       # the HBM roof binds
                    ach, peak, unit, bound = d["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
            else:
                ach, peak, unit, bound = d["work"] / sec / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
            work = d["bytes"] if (d["unit"] == "flop" and bound == "hbm") else d["work"]
            roof_ms = work / (peak * (1e9 if unit == "GB/s" else 1e12)) * 1e3 / steps
            out[name] = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                         "frac": round(ach / peak, 4), "ms_per_step": round(d["ms"] / steps, 3),
                         "roofline_ms_per_step": round(roof_ms, 4),
                         "launches_per_step": d["launches"] / steps,
                         "avg_launch_ms": round(d["ms"] / d["launches"], 4)}
        return out


# ----------------------------------------------------------------------------------------
# CPU baseline: the oracle (a port of the reference formulation) on the host cores
# ----------------------------------------------------------------------------------------
def _measured_traffic(kernel):
    """HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 passes of this same command,
    summarised by tools/traffic_summary.py into profiles/rNN_traffic.json); None if not profiled."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return None if k is None else k["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def _host_cores():
    """Threads for the CPU leg: the cgroup CPU quota if there is one, else the affinity mask,
    never more than 16 (a 1-GPU box's host share; its affinity mask shows the whole machine)."""
    if os.environ.get("MK_CPU_THREADS"):
        return int(os.environ["MK_CPU_THREADS"])
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline():
    """Oracle fwd+bwd on the host cores, on a bounded sample of the step (about 20-30 s of CPU work).

    The full E=384 step takes minutes on a CPU, so each class of work is timed at the true
    grid sizes on a 1/8 .. 1/16 slice along an axis it is exactly linear in (channels for the
    transforms, latitude rows for the pointwise stack, degrees for dhconv) and scaled back:
    """
    import torch.nn as nn
    from oracle import spectral as osp
    ncores = _host_cores()
    torch.set_num_threads(ncores)
    torch.manual_seed(0)
    E, L, M = 384, 240, 241

    def fwd_bwd(fn, *xs):
        xs = [x.requires_grad_(True) for x in xs]
        t0 = time.time()
        y = fn(*xs)
        (y.real if y.is_complex() else y).sum().backward()
        return time.time() - t0

    def note(msg):
        print(f"[cpu_baseline] {msg}", file=sys.stderr, flush=True)

    parts = {}
    # transforms: linear in channels
    for tag, (K, N, grid, cs) in {"full": (721, 1440, "equiangular", 24), "low": (240, 480, "legendre-gauss", 48)}.items():
        f, fi = osp.TorchRealSHT(K, N, L, M, grid), osp.TorchInverseRealSHT(K, N, L, M, grid)
        t_f = fwd_bwd(f, torch.randn(1, cs, K, N)) * (E / cs)
        t_i = fwd_bwd(fi, torch.complex(torch.randn(1, cs, L, M), torch.randn(1, cs, L, M))) * (E / cs)
        parts[f"sht_{tag}"], parts[f"isht_{tag}"] = t_f, t_i
        note(f"{tag}-res SHT {t_f:.2f} s, iSHT {t_i:.2f} s per 384-channel transform (timed on {cs} channels)")
    # dhconv: linear in degrees
    ls = 30
    xs = torch.complex(torch.randn(1, E, ls, M), torch.randn(1, E, ls, M))
    ws = torch.complex(torch.randn(E, E, ls), torch.randn(E, E, ls))
    parts["dhconv"] = fwd_bwd(osp.contract_dhconv, xs, ws) * (L / ls)
    note(f"dhconv {parts['dhconv']:.2f} s per layer (timed on {ls} of {L} degrees)")

    # pointwise stack: linear in latitude rows
    def block_tail(rows, cols):
        norm0, norm1 = nn.InstanceNorm2d(E, eps=1e-6, affine=True), nn.InstanceNorm2d(E, eps=1e-6, affine=True)
        mlp, skip, act = osp.MLP(E, 2 * E), nn.Conv2d(E, E, 1, bias=False), nn.GELU()

        def fn(x, r):
            return norm1(mlp(act(norm0(x)))) + skip(r)
        return fwd_bwd(fn, torch.randn(1, E, rows, cols), torch.randn(1, E, rows, cols))

    parts["block_full"] = block_tail(91, 1440) * (721 / 91)
    parts["block_low"] = block_tail(30, 480) * (240 / 30)
    enc, dec = osp.EncoderDecoder(1, 73, E, E, nn.GELU), osp.EncoderDecoder(1, E, 73, E, nn.GELU, gain=0.5)
    res = nn.Conv2d(73, 73, 1, bias=False)
    parts["enc_dec"] = fwd_bwd(lambda x: dec(enc(x)) + res(x), torch.randn(1, 73, 91, 1440)) * (721 / 91)
    note(f"pointwise: full-res block {parts['block_full']:.2f} s, low-res block {parts['block_low']:.2f} s, "
         f"encoder+decoder {parts['enc_dec']:.2f} s")
    # transform census of one forward: 1 + 7 analyses, 2 + 8 syntheses
    total = (parts["sht_full"] + 2 * parts["isht_full"] + 7 * parts["sht_low"] + 8 * parts["isht_low"]
             + 8 * parts["dhconv"] + parts["block_full"] + 7 * parts["block_low"] + parts["enc_dec"])
    return {"value": round(1.0 / total, 5), "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": ("oracle (fp32 torch-CPU restatement of the reference's rfft+einsum formulation) fwd+bwd, B=1, no "
                       "optimizer: each op class timed at the true grid on a slice it is linear in (transforms on 24 / 48 "
                       "of 384 channels, dhconv on 30 of 240 degrees, pointwise stack on 91 of 721 / 30 of 240 latitude "
                       f"rows), scaled and summed over the step's op census: {total:.1f} s per sample"),
            "parts_s": {k: round(v, 3) for k, v in parts.items()}}


def cpu_full_step():
    """One complete oracle forward + backward of the benchmark configuration (B = 1, fp32) on the host cores: the check of
    the slice extrapolation above (`--full-cpu-baseline`)."""
    from oracle import spectral as osp
    torch.set_num_threads(_host_cores())
    torch.manual_seed(0)
    kw = {k: v for k, v in CONFIG.items() if k not in ("spectral_transform", "filter_type", "pos_embed")}
    net = osp.SphericalFourierNeuralOperatorNet(**kw)
    x, tar = torch.randn(1, 73, 721, 1440), torch.randn(1, 73, 721, 1440)
    t0 = time.time()
    ((net(x) - tar) ** 2).mean().backward()
    dt = time.time() - t0
    print(f"[cpu_baseline] full oracle step: {dt:.1f} s", file=sys.stderr, flush=True)
    return round(dt, 2)


def _synthetic_setup(dev, rank=0, world=1, lat_loc=721, lat_off=0):
    """Weights, fields and row weights of the benchmark (one definition for main(), the golden-loss tool and its test)."""
    from makani_amd import ops
    torch.manual_seed(333 + rank)               # different synthetic data per rank
    B = world
    inp = torch.randn(B, 73, lat_loc, 1440, device=dev)
    tar = torch.randn(B, 73, lat_loc, 1440, device=dev)
    _, wq = ops.quadrature("equiangular", 721)
    wq = torch.from_numpy(wq / wq.sum() / 1440.0).float()[lat_off:lat_off + lat_loc].to(dev)
    return inp, tar, wq.contiguous()


def first_step_losses(with_oracle=False):
    """The loss of the benchmark's first step at N = 1 (bf16 autocast, freshly initialised seed-333 weights), and -- with
    ``with_oracle`` -- the fp32 CPU oracle's loss for the same weights and fields (one oracle forward: ~20 s on 16 cores)."""
    from makani_amd import ops
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(333)
    net = SphericalFourierNeuralOperatorNet(**CONFIG)
    state = {k: v.clone() for k, v in net.state_dict().items()} if with_oracle else None
    net = net.to(dev)
    inp, tar, wq_row = _synthetic_setup(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pred = net(inp)
    out = {"bf16_engine_loss": float(ops.weighted_mse(pred, tar, wq_row, 1.0 / 73).item())}
    if with_oracle:
        from oracle import spectral as osp
        torch.set_num_threads(_host_cores())
        ref = osp.SphericalFourierNeuralOperatorNet(**{k: v for k, v in CONFIG.items()
                                                       if k not in ("spectral_transform", "filter_type", "pos_embed")})
        ref.load_state_dict(state, strict=True)
        with torch.no_grad():
            po = ref(inp.cpu())
        w = wq_row.cpu().double().view(1, 1, -1, 1)
        out["oracle_fp32_loss"] = float((((po.double() - tar.cpu().double()) ** 2) * w).sum().item() / 73)
    return out


def _golden_first_loss():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_first_loss.json")))
    except (OSError, ValueError):
        return None


# ----------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="also run ONE complete oracle forward+backward of the full configuration on the host cores "
                         "(minutes, tens of GB of RAM) and report it next to the slice extrapolation")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward+loss+backward in a HIP graph and replay it (reference: trainer.py:84-152, "
                         "optimizer step outside the graph); kernel timing then comes from an eager pre-pass")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as a torch.distributed.run child BEFORE anything here touches
        # the GPU (no exec of a GPU-initialised process, no HIP context in this parent) and hand its exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from makani_amd import comm, mappings
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd import ops
    spectral_mode = ops.SPECTRAL_GEMM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal hooks (one-GPU box): MK_BENCH_BACKEND=gloo MK_BENCH_ONE_DEVICE=1 run every rank on cuda:0 over gloo,
    # which exercises this script's whole N > 1 path except the RCCL wire.  The driver's runs use neither.
    if world > 1:
        # a collective that stalls (MK_COLLECTIVE_TIMEOUT, default 300 s) aborts the rank with a non-zero exit code
        comm.init(model_parallel_sizes=[world, 1, 1, 1], model_parallel_names=["h", "w", "fin", "fout"],
                  backend=os.environ.get("MK_BENCH_BACKEND", "nccl"))
    rank = comm.get_world_rank()
    local_rank = 0 if os.environ.get("MK_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    hsize, hrank = comm.get_size("h"), comm.get_rank("h")

    torch.manual_seed(333)                      # same replicated weights on every rank ...
    net = SphericalFourierNeuralOperatorNet(**CONFIG).to(dev)
    mappings.sync_params(net)
    from makani_amd.optim import FusedAdam
    # MK_ADAM_OVERLAP=1: the update of a large tensor starts on a side stream as soon as backward has produced its gradient
    # (optim.py).  Measured neutral on one GPU (44.57-44.85 ms/step off, 44.69-44.85 on, same box, alternating runs: the
    # backward kernels running beside an Adam pass slow down by what the pass would have cost on its own), so it stays off.
    adam_overlap = int(os.environ.get("MK_ADAM_OVERLAP", "0"))
    opt = FusedAdam(net.parameters(), lr=1e-4, overlap_backward=adam_overlap)

    B = world                                   # weak scaling: one sample per GPU
    lat_loc = net.inp_shape_loc[0]
    lat_off = sum(net.trans_down.lat_shapes[:hrank]) if hsize > 1 else 0
    inp, tar, wq_row = _synthetic_setup(dev, rank, world, lat_loc, lat_off)    # ... different synthetic data per rank; wq_row:
                                                                               # quadrature weight per local latitude row

    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()

    # MK_BENCH_MICROBATCH=2: two micro-batches on two HIP streams / RCCL communicators, so the distributed-SHT all-to-alls of
    # one overlap the kernels of the other (makani_amd/pipeline.py).  Off by default since round 3: kernels of two streams
    # that share the card are not independent on this hardware (a dense-MFMA workgroup beside an FFT workgroup leaves
    # 16-lane register beats of the latter stale: DESIGN.md section 7.4, profiles/r03_share_stress.txt), so the measured
    # step keeps every kernel alone on its card, as the parity tests do.
    nmb = int(os.environ.get("MK_BENCH_MICROBATCH", "1"))
    runner = None
    if nmb > 1:
        from makani_amd.pipeline import MicroBatchRunner
        assert B % nmb == 0, "the local batch must divide into the micro-batches"
        runner = MicroBatchRunner(nmb)
        mb = B // nmb

        def mb_loss(j):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = net(inp[j * mb:(j + 1) * mb])
            return ops.weighted_mse(pred, tar[j * mb:(j + 1) * mb], wq_row, 1.0 / (B * 73))

    def step():
        opt.zero_grad(set_to_none=True)
        if runner is not None:
            loss = runner.forward(mb_loss)
            loss.backward()
            runner.sync()
            mappings.reduce_shared_gradients(net)
            opt.step()
            return loss
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = net(inp)
        loss = ops.weighted_mse(pred, tar, wq_row, 1.0 / (B * 73))
        loss.backward()
        mappings.reduce_shared_gradients(net)
        opt.step()
        return loss

    # Kernel table: every hand-written spectral / wgrad launch is bracketed with HIP events during the LAST warm-up
    # steps (~500 events per step cost 3 % of the step time, so they are kept out of the timed region); the timed
    # region then records only the launches of the dominant kernel -- the `roofline` figures come from there.
    if runner is not None:
        # one guarded step: an error in the two-stream / two-communicator path is deterministic (every rank runs the same
        # program), so all ranks fall back to the single-stream step together instead of losing the run
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            print(f"[bench] micro-batched step failed ({type(e).__name__}: {e}); continuing on one stream", file=sys.stderr,
                  flush=True)
            runner, nmb = None, 1
            opt.zero_grad(set_to_none=True)

    # value check of the timed configuration: the loss of the very first step (initial weights) against the value recorded in
    # tests/golden/bench_first_loss.json, which tests/test_parity_gpu.py::test_bench_first_step_loss pins to the fp32 CPU oracle
    first_loss = None
    if world == 1 and runner is None and args.warmup > 0:
        first_loss = float(step().item())
        args_warm_left = args.warmup - 1
    else:
        args_warm_left = args.warmup

    timed_step = step
    pre_kernels = None
    n_probe = 0 if (args.no_kernel_timing or args.graph) else min(args_warm_left, 2)
    for i in range(args_warm_left):
        timer.enabled = i >= args_warm_left - n_probe
        step()
    torch.cuda.synchronize()
    timer.enabled = False
    if n_probe:
        pre_kernels = timer.summary(n_probe)
        timer.records.clear()
        timer.only = {max(pre_kernels, key=lambda n: pre_kernels[n]["ms_per_step"])}

    if args.graph:
        if not args.no_kernel_timing:       # events cannot be recorded inside a captured graph
            timer.enabled = True
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            timer.enabled = False
            pre_kernels = timer.summary(2)
            timer.records.clear()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        static_loss = None
        with torch.cuda.graph(graph):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = net(inp)
            static_loss = ops.weighted_mse(pred, tar, wq_row, 1.0 / (B * 73))
            static_loss.backward()

        def timed_step():
            graph.replay()
            mappings.reduce_shared_gradients(net)
            opt.step()
            return static_loss
        timed_step()
        torch.cuda.synchronize()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = not args.no_kernel_timing and not args.graph
    # Python's cyclic garbage collector is kept out of the timed region (a generation-2 pass over the process's objects
    # stalls one step by 70-90 ms around step 10 of a fresh process: tools/step_times.py); collected before and after.
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = timed_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert math.isfinite(loss.item()), "loss is not finite"

    if rank == 0:
        kernels = timer.summary(args.steps) if not args.no_kernel_timing else {}
        timed = dict(kernels)                       # what was recorded inside the timed region
        if pre_kernels is not None:
            kernels = dict(pre_kernels)
            kernels.update(timed)                   # the dominant kernel: timed-region figures
        roof = None
        if kernels:
            dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
            k = kernels[dom]
            roof = {"kernel": dom, "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"], "unit": k["unit"],
                    "frac": k["frac"], "traffic": _measured_traffic(dom), "traffic_unit": "HBM bytes per launch (PMC)",
                    "avg_launch_ms": k["avg_launch_ms"]}
        step_roof = None
        if kernels:
            # step-level roofline: sum over the step's op census of (algorithmic work / the peak that binds it), over the
            # measured step time.  Timed launches bring their own work (KernelTimer); the streaming passes without a
            # wrapper are counted from the configuration: instance norms (block: norm0 + GELU forward = 3 row passes of the
            # bf16 field, norm1's forward rides in GEMM epilogues, each backward = 5), Adam (28 bytes per real parameter:
            # p, g, m, v read, p, m, v written), the loss (pred bf16 + target fp32 read, gradient written: 14 bytes/px)
            E, nblk = CONFIG["embed_dim"], CONFIG["num_layers"]
            px_full = lat_loc * 1440
            px_low = net.h_loc * net.w_loc
            nreal = sum(p.numel() * (2 if p.is_complex() else 1) for p in net.parameters())
            census = {n: k["roofline_ms_per_step"] for n, k in kernels.items() if "roofline_ms_per_step" in k}
            census["instance_norm"] = 13 * E * 2 * B * (px_full + (nblk - 1) * px_low) / (PEAK_HBM_GBS * 1e9) * 1e3
            census["adam"] = 28.0 * nreal / (PEAK_HBM_GBS * 1e9) * 1e3
            census["loss"] = 14.0 * 73 * B * px_full / (PEAK_HBM_GBS * 1e9) * 1e3
            total = sum(census.values())
            step_roof = {"sum_work_over_peak_ms": round(total, 3), "frac": round(total / (1e3 * elapsed / args.steps), 4),
                         "census_ms": {n: round(v, 3) for n, v in census.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
            if args.full_cpu_baseline:
                cpu["full_step_s"] = cpu_full_step()
                cpu["extrapolation_over_full"] = round((1.0 / cpu["value"]) / cpu["full_step_s"], 3)
        line = {
            "metric": "SFNO fwd+bwd samples/sec, 73ch 721x1440",
            "value": round(B * args.steps / elapsed, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "sfno_linear_73chq_sc3_layers8_edim384 fwd+bwd+Adam, 73ch 721x1440, per-GPU batch 1",
                       "global_batch": B, "parallelism": f"h{hsize}", "spectral_dtype": "f32 (GEMMs as 6 bf16 MFMA products of exact 3-way operand splits)"
                       if spectral_mode == "bf16x3" else "f32",
                       "step_launch": "hipGraph replay" if args.graph else "eager",
                       "micro_batches": nmb, "python_gc": "disabled in the timed region",
                       "adam": ("large tensors updated on a side stream as their gradients arrive, joined in opt.step()"
                                if (adam_overlap and not args.graph) else "after backward")},
            "roofline": roof, "step_roofline": step_roof, "cpu_baseline": cpu, "kernels": kernels,
            "kernel_timing": ("HIP events: all kernels over the last %d warm-up step(s), the roofline kernel over the timed region"
                              % n_probe) if n_probe else ("HIP events over an eager pre-pass" if args.graph else
                                                          "HIP events over the timed region"),
            "loss": round(loss.item(), 6),
        }
        golden = _golden_first_loss()
        if first_loss is not None:
            chk = {"first_step_loss": round(first_loss, 6)}
            if golden is not None:
                g = golden["bf16_engine_loss"]
                chk.update(golden=g, oracle_fp32=golden.get("oracle_fp32_loss"), rel_diff=round(abs(first_loss - g) / g, 6),
                           ok=bool(abs(first_loss - g) <= 2e-3 * g))
            line["loss_check"] = chk
        print(json.dumps(line), flush=True)
    if world > 1:
        comm.cleanup()


if __name__ == "__main__":
    main()
