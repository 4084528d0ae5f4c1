"""Two ranks (h = 2) sharing cuda:0, gloo transport, REAL HIP kernels: the distributed SHT, the
instance norm and the full SFNO step against the serial CPU oracle.  Complements the CPU gloo tests
(which stub the local kernels) -- here only the wire (gloo instead of RCCL) differs from production.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rel(a, b, floor=0.0):
    a, b = a.detach().cpu(), b.detach().cpu()
    cplx = a.is_complex() or b.is_complex()
    a = a.to(torch.complex128) if cplx else a.double()
    b = b.to(torch.complex128) if cplx else b.double()
    return (torch.linalg.norm(a - b) / max(torch.linalg.norm(b).item(), floor)).item()


def _gather(x, dim, name):
    from makani_amd import comm
    size = comm.get_size(name)
    if size == 1:
        return x
    sizes = [torch.zeros(1, dtype=torch.long) for _ in range(size)]
    dist.all_gather(sizes, torch.tensor([x.shape[dim]]), group=comm.get_group(name))
    sizes = [int(s) for s in sizes]
    pad = max(sizes) - x.shape[dim]
    xp = x.contiguous()
    if pad:
        shp = list(x.shape)
        shp[dim] = pad
        xp = torch.cat([xp, torch.zeros(shp, dtype=x.dtype, device=x.device)], dim=dim)
    outs = [torch.empty_like(xp) for _ in range(size)]
    dist.all_gather(outs, xp, group=comm.get_group(name))
    return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)


def _shard(x, dim, name):
    from makani_amd import comm
    from makani_amd.distributed import split_tensor_along_dim
    if comm.get_size(name) == 1:
        return x
    return split_tensor_along_dim(x, dim, comm.get_size(name))[comm.get_rank(name)].contiguous()


def _body_sht(dev):
    from makani_amd.distributed import DistributedRealSHT, DistributedInverseRealSHT
    from oracle import spectral as osp
    # planned FFT kernels / split kernels / split kernels with channel blocks of 24 (peer-major, copy-free exchange)
    for nlat, nlon, lmax, mmax, B, C in ((91, 180, 30, 31, 2, 6), (33, 480, 32, 33, 1, 8), (33, 480, 32, 33, 2, 48)):
        _sht_case(dev, nlat, nlon, lmax, mmax, B, C)


def _sht_case(dev, nlat, nlon, lmax, mmax, B, C):
    from makani_amd.distributed import DistributedRealSHT, DistributedInverseRealSHT
    from oracle import spectral as osp
    torch.manual_seed(333)
    f = DistributedRealSHT(nlat, nlon, lmax, mmax, "equiangular").to(dev)
    fi = DistributedInverseRealSHT(nlat, nlon, lmax, mmax, "equiangular").to(dev)
    fo, fio = osp.TorchRealSHT(nlat, nlon, lmax, mmax, "equiangular"), osp.TorchInverseRealSHT(nlat, nlon, lmax, mmax, "equiangular")
    xg = torch.randn(B, C, nlat, nlon)
    gg = torch.complex(torch.randn(B, C, lmax, mmax), torch.randn(B, C, lmax, mmax))
    xo = xg.clone().requires_grad_(True)
    co = fo(xo)
    co.backward(gg)
    xl = _shard(xg, 2, "h").to(dev).requires_grad_(True)
    cl = f(xl)
    cl.backward(_shard(gg, 2, "h").to(dev))
    assert _rel(_gather(cl.detach(), 2, "h"), co.detach()) < 1e-5
    assert _rel(_gather(xl.grad, 2, "h"), xo.grad) < 1e-5
    yo = fio(co.detach())
    yl = fi(_shard(co.detach(), 2, "h").to(dev))
    assert _rel(_gather(yl, 2, "h"), yo) < 1e-5


def _body_norm(dev):
    """DistributedInstanceNorm2d on the HIP kernels (local sums -> all-reduce -> apply) vs the serial torch norm,
    uneven latitude shards, with and without the fused GELU, including the shared weight / bias gradients."""
    from makani_amd import mappings
    from makani_amd.layer_norm import DistributedInstanceNorm2d
    torch.manual_seed(5)
    B, C, H, W = 2, 6, 33, 64
    xg, gg = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
    wg, bg = torch.rand(C) + 0.5, torch.randn(C)
    for fuse in (False, True):
        ref = torch.nn.InstanceNorm2d(C, eps=1e-6, affine=True)
        with torch.no_grad():
            ref.weight.copy_(wg)
            ref.bias.copy_(bg)
        xo = xg.clone().requires_grad_(True)
        yo = ref(xo)
        if fuse:
            yo = torch.nn.functional.gelu(yo)
        yo.backward(gg)
        mod = DistributedInstanceNorm2d(C, eps=1e-6, affine=True).to(dev)
        with torch.no_grad():
            mod.weight.copy_(wg)
            mod.bias.copy_(bg)
        xl = _shard(xg, 2, "h").to(dev).requires_grad_(True)
        yl = mod(xl, fuse_gelu=fuse)
        yl.backward(_shard(gg, 2, "h").to(dev))
        mappings.reduce_shared_gradients(mod)
        assert _rel(_gather(yl.detach(), 2, "h"), yo.detach()) < 1e-5
        assert _rel(_gather(xl.grad, 2, "h"), xo.grad) < 1e-5
        assert _rel(mod.weight.grad, ref.weight.grad) < 1e-5
        assert _rel(mod.bias.grad, ref.bias.grad) < 1e-5


def _body_net(dev):
    from makani_amd import comm, mappings
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd.distributed import compute_split_shapes
    from oracle import spectral as osp
    torch.manual_seed(333)
    kw = dict(inp_shape=(64, 128), out_shape=(64, 128), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw)
    hs, hr = comm.get_size("h"), comm.get_rank("h")
    sd = ref.state_dict()
    for k in list(sd):
        if k.endswith("filter.filter.weight"):
            sd[k] = torch.split(sd[k], compute_split_shapes(sd[k].shape[-1], hs), dim=-1)[hr].contiguous()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    B = 2
    xg, tg = torch.randn(B, 4, 64, 128), torch.randn(B, 3, 64, 128)
    xo = xg.clone().requires_grad_(True)
    yo = ref(xo)
    ((yo - tg) ** 2).sum().backward()
    xl = _shard(xg, 2, "h").to(dev).requires_grad_(True)
    yl = net(xl)
    ((yl - _shard(tg, 2, "h").to(dev)) ** 2).sum().backward()
    mappings.reduce_shared_gradients(net)
    assert _rel(_gather(yl.detach(), 2, "h"), yo.detach()) < 2e-5
    assert _rel(_gather(xl.grad, 2, "h"), xo.grad) < 5e-5
    po = dict(ref.named_parameters())
    scale = float(np.median([p.grad.norm().item() for p in po.values()]))
    for n, p in net.named_parameters():
        want = po[n].grad
        if n.endswith("filter.filter.weight"):
            want = torch.split(want, compute_split_shapes(want.shape[-1], hs), dim=-1)[hr]
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert _rel(got, want, floor=0.1 * scale) < 1e-4, n


def _body_pipeline(dev):
    """The SFNO step as two overlapped micro-batches (two streams, two comm lanes) on h = 2 vs the serial oracle."""
    from makani_amd import comm, mappings
    from makani_amd.pipeline import MicroBatchRunner
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd.distributed import compute_split_shapes
    from oracle import spectral as osp
    torch.manual_seed(333)
    kw = dict(inp_shape=(64, 128), out_shape=(64, 128), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw)
    hs, hr = comm.get_size("h"), comm.get_rank("h")
    sd = ref.state_dict()
    for k in list(sd):
        if k.endswith("filter.filter.weight"):
            sd[k] = torch.split(sd[k], compute_split_shapes(sd[k].shape[-1], hs), dim=-1)[hr].contiguous()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    B = 2
    xg, tg = torch.randn(B, 4, 64, 128), torch.randn(B, 3, 64, 128)
    yo = ref(xg)
    ((yo - tg) ** 2).sum().backward()
    xl, tl = _shard(xg, 2, "h").to(dev), _shard(tg, 2, "h").to(dev)
    runner = MicroBatchRunner(2)
    assert comm.num_lanes() == 2
    preds = [None, None]

    def mb_loss(j):
        preds[j] = net(xl[j:j + 1])
        return ((preds[j] - tl[j:j + 1]) ** 2).sum()

    runner.forward(mb_loss).backward()
    runner.sync()
    mappings.reduce_shared_gradients(net)
    yl = torch.cat([p.detach() for p in preds], dim=0)
    assert _rel(_gather(yl, 2, "h"), yo.detach()) < 2e-5
    po = dict(ref.named_parameters())
    scale = float(np.median([p.grad.norm().item() for p in po.values()]))
    for n, p in net.named_parameters():
        want = po[n].grad
        if n.endswith("filter.filter.weight"):
            want = torch.split(want, compute_split_shapes(want.shape[-1], hs), dim=-1)[hr]
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert _rel(got, want, floor=0.1 * scale) < 1e-4, n


def _body_net_bf16(dev):
    """h = 2 under bf16 autocast: the pointwise stack runs on the pixel-column engine, the block tail on the fused
    conv + norm node (local row sums -> all-reduce over the spatial group -> coefficients -> epilogue).  Against the fp32
    serial oracle at bf16 accuracy; the fused tail against the unfused one (MK_NORM_SKIP_FUSION=0) at rounding level."""
    from makani_amd import comm, mappings
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd.distributed import compute_split_shapes
    from oracle import spectral as osp
    torch.manual_seed(77)
    kw = dict(inp_shape=(64, 128), out_shape=(64, 128), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=16, num_layers=2)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw)
    hs, hr = comm.get_size("h"), comm.get_rank("h")
    sd = ref.state_dict()
    for k in list(sd):
        if k.endswith("filter.filter.weight"):
            sd[k] = torch.split(sd[k], compute_split_shapes(sd[k].shape[-1], hs), dim=-1)[hr].contiguous()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    xg, tg = torch.randn(2, 4, 64, 128), torch.randn(2, 3, 64, 128)
    yo = ref(xg)
    ((yo - tg) ** 2).sum().backward()
    xl, tl = _shard(xg, 2, "h").to(dev), _shard(tg, 2, "h").to(dev)

    def step():
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(xl)
        ((y.float() - tl) ** 2).sum().backward()
        mappings.reduce_shared_gradients(net)
        return y.detach().float(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    y1, g1 = step()
    assert _rel(_gather(y1, 2, "h"), yo.detach()) < 3e-2
    po = dict(ref.named_parameters())
    scale = float(np.median([p.grad.norm().item() for p in po.values()]))
    for n, g in g1.items():
        want = po[n].grad
        if n.endswith("filter.filter.weight"):
            want = torch.split(want, compute_split_shapes(want.shape[-1], hs), dim=-1)[hr]
        assert _rel(g, want, floor=0.1 * scale) < 8e-2, n
    os.environ["MK_NORM_SKIP_FUSION"] = "0"
    try:
        y0, g0 = step()
    finally:
        del os.environ["MK_NORM_SKIP_FUSION"]
    assert _rel(y1, y0) < 1e-2
    for n in g1:
        assert _rel(g1[n], g0[n], floor=0.1 * scale) < 2e-2, n


def _body_sht_w(dev):
    """Longitude sharding (w = 2): the azimuth transposes around the FFT with the real kernels."""
    from makani_amd.distributed import DistributedRealSHT, DistributedInverseRealSHT
    from oracle import spectral as osp
    torch.manual_seed(12)
    nlat, nlon, lmax, mmax, B, C = 33, 480, 32, 33, 2, 6
    f = DistributedRealSHT(nlat, nlon, lmax, mmax, "equiangular").to(dev)
    fi = DistributedInverseRealSHT(nlat, nlon, lmax, mmax, "equiangular").to(dev)
    fo, fio = osp.TorchRealSHT(nlat, nlon, lmax, mmax, "equiangular"), osp.TorchInverseRealSHT(nlat, nlon, lmax, mmax, "equiangular")
    xg = torch.randn(B, C, nlat, nlon)
    gg = torch.complex(torch.randn(B, C, lmax, mmax), torch.randn(B, C, lmax, mmax))
    xo = xg.clone().requires_grad_(True)
    co = fo(xo)
    co.backward(gg)
    xl = _shard(xg, 3, "w").to(dev).requires_grad_(True)
    cl = f(xl)
    cl.backward(_shard(gg, 3, "w").to(dev))
    assert _rel(_gather(cl.detach(), 3, "w"), co.detach()) < 1e-5
    assert _rel(_gather(xl.grad, 3, "w"), xo.grad) < 1e-5
    yl = fi(_shard(co.detach(), 3, "w").to(dev))
    assert _rel(_gather(yl, 3, "w"), fio(co.detach())) < 1e-5


def _bcast_from0(t, shape, dtype):
    """Tensor computed on rank 0 (CPU) to every rank (gloo)."""
    buf = t.contiguous() if dist.get_rank() == 0 else torch.empty(shape, dtype=dtype)
    if buf.is_complex():
        dist.broadcast(torch.view_as_real(buf), src=0)
    else:
        dist.broadcast(buf, src=0)
    return buf


def _body_sht_prod(dev):
    """BASELINE configs[3] shard shapes on the REAL kernels: the production transforms (721x1440 equiangular -> 240 x 241 modes and
    the 240x480 Legendre-Gauss pair), 384 channels, latitude shards 721 -> [181, 181, 181, 178] / 240 -> [60] x 4, 96 channels
    per rank (multiples of 24: peer-major Fourier rows, the copy-free exchange): forward value, input gradient and the
    synthesis against the serial fp32 oracle (computed on rank 0, broadcast)."""
    from makani_amd import comm
    from makani_amd.distributed import DistributedRealSHT, DistributedInverseRealSHT
    from oracle import spectral as osp
    L, M, B, C = 240, 241, 1, 384
    for (nlat, nlon, grid) in ((721, 1440, "equiangular"), (240, 480, "legendre-gauss")):
        torch.manual_seed(333)
        f = DistributedRealSHT(nlat, nlon, L, M, grid).to(dev)
        fi = DistributedInverseRealSHT(nlat, nlon, L, M, grid).to(dev)
        assert f.lat_shapes == ([181, 181, 181, 178] if nlat == 721 else [60] * 4) and f.l_shapes == [60] * 4
        xg = torch.randn(B, C, nlat, nlon)
        gg = torch.complex(torch.randn(B, C, L, M), torch.randn(B, C, L, M))
        if comm.get_world_rank() == 0:
            torch.set_num_threads(16)
            fo, fio = osp.TorchRealSHT(nlat, nlon, L, M, grid), osp.TorchInverseRealSHT(nlat, nlon, L, M, grid)
            xo = xg.clone().requires_grad_(True)
            co = fo(xo)
            co.backward(gg)
            co, gxo = co.detach(), xo.grad
            yo = fio(co)
        else:
            co = gxo = yo = None
        co = _bcast_from0(co, (B, C, L, M), torch.complex64)
        gxo = _bcast_from0(gxo, (B, C, nlat, nlon), torch.float32)
        yo = _bcast_from0(yo, (B, C, nlat, nlon), torch.float32)
        xl = _shard(xg, 2, "h").to(dev).requires_grad_(True)
        cl = f(xl)
        cl.backward(_shard(gg, 2, "h").to(dev))
        assert _rel(cl.detach(), _shard(co, 2, "h")) < 1e-5
        assert _rel(xl.grad, _shard(gxo, 2, "h")) < 1e-5
        yl = fi(_shard(co, 2, "h").to(dev))
        assert _rel(yl, _shard(yo, 2, "h")) < 1e-5
        del f, fi, xl, cl, yl
        torch.cuda.empty_cache()


def _body_net_prod(dev):
    """configs[3]: sfno_linear_73chq_sc3_layers8_edim384 forward AND backward with h_parallel_size = 4 on the real kernels (fp32):
    every rank's latitude shard of the output and of the input gradient, and a sample of parameter gradients (shared weights after
    the SUM over `h` of `mappings.reduce_shared_gradients`, the l-sharded spectral weight shard by shard), against ONE oracle step
    on rank 0 (broadcast).  The loss is `sum(y * g)` with a fixed random field g, so the local losses add up to the serial one."""
    import bench
    from makani_amd import comm, mappings
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd.distributed import compute_split_shapes
    from oracle import spectral as osp
    torch.manual_seed(333)
    ref = osp.SphericalFourierNeuralOperatorNet(**{k: v for k, v in bench.CONFIG.items()
                                                   if k not in ("spectral_transform", "filter_type", "pos_embed")})
    net = SphericalFourierNeuralOperatorNet(**bench.CONFIG)
    hs, hr = comm.get_size("h"), comm.get_rank("h")
    sd = ref.state_dict()
    for k in list(sd):
        if k.endswith("filter.filter.weight"):
            sd[k] = torch.split(sd[k], compute_split_shapes(sd[k].shape[-1], hs), dim=-1)[hr].contiguous()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    assert net.inp_shape_loc[0] == [181, 181, 181, 178][hr]
    xg = torch.randn(1, 73, 721, 1440)
    gg = torch.randn(1, 73, 721, 1440)
    shared = ["encoder.fwd.0.weight", "blocks.3.mlp.fwd.0.weight", "blocks.7.norm0.weight", "decoder.fwd.2.weight",
              "residual_transform.weight"]
    sharded = "blocks.0.filter.filter.weight"
    shapes = {n: tuple(p.shape) for n, p in ref.named_parameters()}
    if comm.get_world_rank() == 0:
        torch.set_num_threads(16)
        xo = xg.clone().requires_grad_(True)
        yo = ref(xo)
        (yo * gg).sum().backward()
        yo, gxo = yo.detach(), xo.grad
        po = {n: p.grad for n, p in ref.named_parameters() if n in shared or n == sharded}
    else:
        yo = gxo = None
        po = {}
    del ref
    yo = _bcast_from0(yo, (1, 73, 721, 1440), torch.float32)
    gxo = _bcast_from0(gxo, (1, 73, 721, 1440), torch.float32)
    want = {n: _bcast_from0(po.get(n), shapes[n], torch.complex64 if n == sharded else torch.float32) for n in shared + [sharded]}
    xl = _shard(xg, 2, "h").to(dev).requires_grad_(True)
    yl = net(xl)
    (yl * _shard(gg, 2, "h").to(dev)).sum().backward()
    mappings.reduce_shared_gradients(net)
    err = _rel(yl, _shard(yo, 2, "h"))
    assert err < 2e-5, f"rank {hr}: output error {err:.3e}"
    err = _rel(xl.grad, _shard(gxo, 2, "h"))
    assert err < 5e-5, f"rank {hr}: input gradient error {err:.3e}"
    pn = dict(net.named_parameters())
    scale = float(np.median([torch.linalg.norm(want[n].to(torch.complex128) if want[n].is_complex() else want[n].double()).item()
                             for n in shared]))
    for n in shared:
        err = _rel(pn[n].grad, want[n], floor=1e-1 * scale)
        assert err < 5e-5, f"rank {hr}: gradient of {n}: {err:.3e}"
    w = torch.split(want[sharded], compute_split_shapes(want[sharded].shape[-1], hs), dim=-1)[hr]
    err = _rel(pn[sharded].grad, w)
    assert err < 5e-5, f"rank {hr}: gradient of the l-sharded {sharded}: {err:.3e}"

    # the same step in the benchmark's mode -- bf16 autocast: pixel-column engine, per-step arena, the inverse FFT's row statistics
    # through the sharded instance norm's all-reduce -- against the same fp32 oracle at bf16 tolerances (those of the
    # single-GPU test, tests/test_parity_gpu.py: output 3e-2, input gradient 6e-2, parameter gradients 8e-2)
    net.zero_grad(set_to_none=True)
    xb = _shard(xg, 2, "h").to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = net(xb)
    (yb.float() * _shard(gg, 2, "h").to(dev)).sum().backward()
    mappings.reduce_shared_gradients(net)
    e_y, e_x = _rel(yb.float(), _shard(yo, 2, "h")), _rel(xb.grad, _shard(gxo, 2, "h"))
    assert e_y < 3e-2 and e_x < 6e-2, f"rank {hr} (bf16): output {e_y:.3e}, input gradient {e_x:.3e}"
    for n in shared:
        err = _rel(pn[n].grad, want[n], floor=1e-1 * scale)
        assert err < 8e-2, f"rank {hr} (bf16): gradient of {n}: {err:.3e}"
    err = _rel(pn[sharded].grad, w)
    assert err < 8e-2, f"rank {hr} (bf16): gradient of the l-sharded {sharded}: {err:.3e}"


def _take_turns_on_the_card(lock):
    """The ranks of these tests share ONE card, which production never does (one process per GPU).  Kernels of different
    processes (or streams) that run at the same moment are not independent on this hardware: a workgroup of dense bf16 MFMAs
    beside an LDS-exchange kernel (the FFT rows here; rocFFT's just the same) leaves single 16-lane register beats of the
    latter stale -- tools/ab/share_stress.py, DESIGN.md section 7.4.  So a rank computes only while it holds ``lock`` and hands
    it over, with its stream drained, around every collective: what the test then exercises is what production runs, a
    rank's kernels alone on its card, with gloo in place of RCCL on the wire."""
    def wrap(fn):
        def call(*args, **kwargs):
            torch.cuda.synchronize()
            lock.release()
            try:
                return fn(*args, **kwargs)
            finally:
                torch.cuda.synchronize()          # gloo's device copies
                lock.acquire()
        return call
    for name in ("all_to_all_single", "all_reduce", "all_gather", "broadcast", "barrier", "all_to_all", "reduce_scatter_tensor",
                 "all_gather_into_tensor", "new_group", "all_gather_object", "broadcast_object_list", "reduce"):   # every blocking call
        setattr(dist, name, wrap(getattr(dist, name)))


def _worker(rank, world, port, what, q, lock):
    held = False
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0")
        from makani_amd import comm
        comm.init(model_parallel_sizes=[1, world, 1, 1] if what.endswith("_w") else [world, 1, 1, 1], backend="gloo")
        dev = torch.device("cuda:0")
        _take_turns_on_the_card(lock)
        lock.acquire()
        held = True
        globals()["_body_" + what](dev)
        torch.cuda.synchronize()
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        if held:
            try:
                lock.release()
            except ValueError:
                pass
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("what", ["sht", "sht_w", "norm", "net", "net_bf16", "pipeline"])
def test_h2_on_one_gpu(what):
    assert torch.cuda.device_count() >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    lock = ctx.Lock()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, what, q, lock)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in results if r[1] != "ok"]
    assert not bad, "\n".join(f"rank {r}: {m}" for r, m in bad)


@pytest.mark.parametrize("what", ["sht_prod", "net_prod"])
def test_h4_production_shards_on_one_gpu(what):
    """BASELINE configs[3] (h_parallel_size = 4) at its own sizes: four ranks sharing cuda:0, gloo wire, real kernels."""
    assert torch.cuda.device_count() >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    lock = ctx.Lock()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, what, q, lock)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in results if r[1] != "ok"]
    assert not bad, "\n".join(f"rank {r}: {m}" for r, m in bad)


def test_h8_shard_shapes_single_process():
    """BASELINE configs[4] shard shapes (h_parallel_size = 8: 721 -> [91] x 7 + [84] latitudes, 240 -> [30] x 8 degrees, 48 channels
    per rank) on the real kernels.  A GPU box admits at most six processes on its card, so the eight ranks are walked ONE AFTER
    THE OTHER in this process: every rank's local stages are the package's raw ops in the order of
    `DistributedRealSHT.forward_packed` / `DistributedInverseRealSHT.inverse_packed` (makani_amd/distributed.py:169-243: peer-major
    FFT rows, latitude-major Legendre), the two all-to-alls are index shuffles on the device; against the serial fp32 oracle."""
    from makani_amd import ops
    from makani_amd.distributed import compute_split_shapes
    from makani_amd.sht import RealSHT, InverseRealSHT
    from oracle import spectral as osp
    dev = torch.device("cuda:0")
    h, L, M, B, C = 8, 240, 241, 1, 384
    Ch = C // h
    for (nlat, nlon, grid) in ((721, 1440, "equiangular"), (240, 480, "legendre-gauss")):
        torch.manual_seed(333)
        lat, ls = compute_split_shapes(nlat, h), compute_split_shapes(L, h)
        assert lat == ([91] * 7 + [84] if nlat == 721 else [30] * 8) and ls == [30] * 8
        assert ops.fft_pm_supported(nlon, M, C, Ch)
        ser, seri = RealSHT(nlat, nlon, L, M, grid).to(dev), InverseRealSHT(nlat, nlon, L, M, grid).to(dev)    # tables + twiddles
        xg = torch.randn(B, C, nlat, nlon)
        torch.set_num_threads(16)
        with torch.no_grad():
            co = osp.TorchRealSHT(nlat, nlon, L, M, grid)(xg)
            yo = osp.TorchInverseRealSHT(nlat, nlon, L, M, grid)(co)
        xs = [t.contiguous().to(dev) for t in torch.split(xg, lat, dim=2)]
        with torch.no_grad():
            # ---- analysis: FFT on every rank's latitudes, peer-major rows [h, K_loc, M, B * C / h]
            xf = [ops.rfft_pm_raw(x.reshape(B * C, x.shape[2], nlon), ser.twiddles, M, *_scales(nlon), C, Ch) for x in xs]
            spec = []
            for r in range(h):
                # all-to-all 1: rank r receives block r of every peer, latitudes in rank order -> [K, M, B * C / h]
                rows = torch.cat([xf[p][r] for p in range(h)], dim=0).contiguous()
                spec.append(ops.legendre_fwd_raw(rows.view(nlat, M, B * Ch), ser.weights, L, 0, kmajor=True))     # [L, M, B * Ch]
            for r in range(h):
                # all-to-all 2: rank r keeps degrees l_r of every peer's channels -> [l_loc, M, B, C]
                l0 = sum(ls[:r])
                c_r = torch.cat([spec[p][l0:l0 + ls[r]].view(ls[r], M, B, Ch) for p in range(h)], dim=3)
                got = ops.spec_unpack_raw(c_r.reshape(ls[r], M, B * C).contiguous(), l0, 0).view(B, C, ls[r], M)
                assert _rel(got, co[:, :, l0:l0 + ls[r]]) < 1e-5, (nlat, r)
            # ---- synthesis from the oracle's coefficients, the same walk backwards
            cd = co.to(dev)
            cp = [ops.spec_pack_raw(cd[:, :, sum(ls[:r]):sum(ls[:r + 1])].reshape(B * C, ls[r], M).contiguous()).view(ls[r], M, B, C)
                  for r in range(h)]
            xfi = []
            for r in range(h):
                # all-to-all 1 (reverse): rank r gets its channel block of every degree -> [L, M, B * C / h]
                c_all = torch.cat([cp[p][..., r * Ch:(r + 1) * Ch] for p in range(h)], dim=0).contiguous()
                xfi.append(ops.legendre_inv_raw(c_all.view(L, M, B * Ch), seri.pct, nlat, 0, kmajor=True))           # [K, M, B * Ch]
            for r in range(h):
                # all-to-all 2 (reverse): rank r receives its latitudes from every peer, peer-major -> [h, K_loc, M, B * Ch]
                k0 = sum(lat[:r])
                rows = torch.stack([xfi[p][k0:k0 + lat[r]] for p in range(h)], dim=0).contiguous()
                y = ops.irfft_pm_raw(rows, seri.twiddles, nlon, 1.0, 1.0, 1.0, C, Ch).view(B, C, lat[r], nlon)
                assert _rel(y, yo[:, :, k0:k0 + lat[r]]) < 1e-5, (nlat, r)
        del ser, seri, xs, xf, spec, cp, xfi
        torch.cuda.empty_cache()


def _scales(nlon):
    import math
    s = 2.0 * math.pi / nlon
    return s, s, s
