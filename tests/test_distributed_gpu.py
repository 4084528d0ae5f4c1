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


def _worker(rank, world, port, what, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0")
        from makani_amd import comm
        comm.init(model_parallel_sizes=[1, world, 1, 1] if what.endswith("_w") else [world, 1, 1, 1], backend="gloo")
        dev = torch.device("cuda:0")
        globals()["_body_" + what](dev)
        torch.cuda.synchronize()
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("what", ["sht", "sht_w", "norm", "net", "net_bf16", "pipeline"])
def test_h2_on_one_gpu(what):
    assert torch.cuda.device_count() >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, what, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in results if r[1] != "ok"]
    assert not bad, "\n".join(f"rank {r}: {m}" for r, m in bad)
