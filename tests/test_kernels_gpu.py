"""GPU parity tests of the HIP kernels, called through the C ABI (makani_amd.ops -> ctypes).

Checker: the CPU oracle (oracle/), numpy FFTs and explicit einsums on the same seeded
inputs.  Tolerance: relative L2 <= 1e-5 in fp32 (BASELINE.json north_star); the
kernels are exact-fp32 MFMA / VALU so the observed error is ~1e-6 or better.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from makani_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def tril_mask(lmax, mmax, l_off=0, m_off=0):
    l = np.arange(lmax)[:, None] + l_off
    m = np.arange(mmax)[None, :] + m_off
    return (l >= m)


# --------------------------------------------------------------------------- FFT
FFT_CASES = [  # bc, nlat, nlon, mmax
    (5, 7, 1440, 241),     # production full-res length, odd channel count (ragged tile)
    (19, 3, 480, 241),     # production low-res length incl. the Nyquist mode
    (3, 33, 64, 17),
    (16, 2, 180, 31),
    (4, 5, 36, 19),        # generic (direct DFT) path, all modes
    (2, 3, 720, 100),
    (1, 1, 16, 9),
]


@pytest.mark.parametrize("bc,nlat,nlon,mmax", FFT_CASES)
def test_rfft_matches_numpy(dev, bc, nlat, nlon, mmax):
    from makani_amd import ops
    rng = np.random.default_rng(333)
    x = rng.standard_normal((bc, nlat, nlon)).astype(np.float32)
    tw = ops.fft_twiddles(nlon).to(dev)
    s = 2 * math.pi / nlon
    xf = ops.rfft_raw(torch.from_numpy(x).to(dev), tw, mmax, s, s, s)
    want = (2 * np.pi * np.fft.rfft(x.astype(np.float64), axis=-1, norm="forward"))[..., :mmax]
    got = xf.cpu().numpy().transpose(2, 1, 0)  # [M,K,BC] -> [BC,K,M]
    assert rel(got, want) < TOL
    # bf16 input rows: same values after an exact up-cast
    xb = torch.from_numpy(x).to(dev).to(torch.bfloat16)
    xfb = ops.rfft_raw(xb, tw, mmax, s, s, s)
    wantb = (2 * np.pi * np.fft.rfft(xb.float().cpu().numpy().astype(np.float64), axis=-1, norm="forward"))[..., :mmax]
    assert rel(xfb.cpu().numpy().transpose(2, 1, 0), wantb) < TOL


@pytest.mark.parametrize("bc,nlat,nlon,mmax", FFT_CASES)
def test_irfft_matches_numpy(dev, bc, nlat, nlon, mmax):
    from makani_amd import ops
    rng = np.random.default_rng(334)
    X = (rng.standard_normal((bc, nlat, mmax)) + 1j * rng.standard_normal((bc, nlat, mmax))).astype(np.complex64)
    tw = ops.fft_twiddles(nlon).to(dev)
    xf = torch.from_numpy(np.ascontiguousarray(X.transpose(2, 1, 0))).to(dev)
    x = ops.irfft_raw(xf, tw, nlon, 1.0, 1.0, 1.0)
    want = np.fft.irfft(X.astype(np.complex128), n=nlon, axis=-1, norm="forward")
    assert rel(x.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("bc,nlat,nlon,mmax", FFT_CASES[:5])
def test_fft_adjoint_pairs(dev, bc, nlat, nlon, mmax):
    """<rfft(x), G> == <x, rfft^H G> and the same for irfft, in torch's gradient convention."""
    from makani_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    tw = ops.fft_twiddles(nlon).to(dev)
    x = torch.randn(bc, nlat, nlon, generator=g).to(dev).requires_grad_(True)
    G = torch.complex(torch.randn(mmax, nlat, bc, generator=g), torch.randn(mmax, nlat, bc, generator=g)).to(dev)
    y = ops.rfft(x, tw, mmax)
    (gx,) = torch.autograd.grad(y, x, G)
    xr = torch.fft.rfft(x.detach().cpu().double(), dim=-1, norm="forward")[..., :mmax] * 2 * math.pi
    xr.requires_grad_(False)
    xc = x.detach().cpu().double().requires_grad_(True)
    yr = (torch.fft.rfft(xc, dim=-1, norm="forward")[..., :mmax] * 2 * math.pi).permute(2, 1, 0)
    (gref,) = torch.autograd.grad(yr, xc, G.cpu().to(torch.complex128))
    assert rel(gx.cpu().numpy(), gref.numpy()) < TOL

    c = torch.complex(torch.randn(mmax, nlat, bc, generator=g), torch.randn(mmax, nlat, bc, generator=g)).to(dev)
    c.requires_grad_(True)
    gy = torch.randn(bc, nlat, nlon, generator=g).to(dev)
    z = ops.irfft(c, tw, nlon)
    (gc,) = torch.autograd.grad(z, c, gy)
    cc = c.detach().cpu().to(torch.complex128).requires_grad_(True)
    zr = torch.fft.irfft(cc.permute(2, 1, 0), n=nlon, dim=-1, norm="forward")
    (gcref,) = torch.autograd.grad(zr, cc, gy.cpu().double())
    assert rel(gc.cpu().numpy(), gcref.numpy()) < TOL


# --------------------------------------------------------------------------- Legendre
LEG_CASES = [  # grid, nlat, lmax, mmax, bc
    ("equiangular", 33, 16, 17, 3),
    ("legendre-gauss", 32, 32, 33, 4),
    ("equiangular", 91, 30, 31, 7),       # odd 2*bc/2 -> float2 staging path
    ("legendre-gauss", 240, 240, 241, 6),  # production low-res
    ("equiangular", 721, 240, 241, 4),     # production full-res
    ("equiangular", 65, 70, 40, 66),       # lmax > TM tiles, several column tiles
]


GEMM_MODES = ["bf16x3", "f32"]   # bf16 MFMA with exact 3-way operand split (default) / fp32 MFMA


@pytest.mark.parametrize("mode", GEMM_MODES)
@pytest.mark.parametrize("grid,nlat,lmax,mmax,bc", LEG_CASES)
def test_legendre_fwd_inv(dev, grid, nlat, lmax, mmax, bc, mode):
    from makani_amd import ops
    rng = np.random.default_rng(7)
    tabw = ops.legendre_table(grid, nlat, lmax, mmax, True)
    tabp = ops.legendre_table(grid, nlat, lmax, mmax, False)
    xf = (rng.standard_normal((mmax, nlat, bc)) + 1j * rng.standard_normal((mmax, nlat, bc))).astype(np.complex64)
    c = ops.legendre_fwd_raw(torch.from_numpy(xf).to(dev), tabw.to(dev), lmax, mode=mode)
    W = tabw.numpy()[:, :, :nlat].astype(np.float64)
    want = np.einsum("mlk,mkn->lmn", W, xf.astype(np.complex128))
    mask = tril_mask(lmax, mmax)[:, :, None]
    got = np.where(mask, c.cpu().numpy(), 0)
    assert rel(got, want) < TOL
    # synthesis: entries with l < m must be ignored whatever they hold
    cin = (rng.standard_normal((lmax, mmax, bc)) + 1j * rng.standard_normal((lmax, mmax, bc))).astype(np.complex64)
    cin = np.where(mask, cin, np.complex64(complex(np.nan, np.nan)))
    y = ops.legendre_inv_raw(torch.from_numpy(cin).to(dev), tabp.to(dev), nlat, mode=mode)
    P = tabp.numpy()[:, :, :nlat].astype(np.float64)
    wanty = np.einsum("mlk,lmn->mkn", P, np.where(mask, cin, 0).astype(np.complex128))
    assert rel(y.cpu().numpy(), wanty) < TOL


@pytest.mark.parametrize("mode", GEMM_MODES)
def test_legendre_mode_shard(dev, mode):
    """m-sharded launch (w-parallel tables): local modes [m_off, m_off + mloc)."""
    from makani_amd import ops
    rng = np.random.default_rng(8)
    grid, nlat, lmax, mmax, bc = "equiangular", 49, 24, 25, 5
    tabw = ops.legendre_table(grid, nlat, lmax, mmax, True).to(dev)
    tabp = ops.legendre_table(grid, nlat, lmax, mmax, False).to(dev)
    xf = torch.from_numpy((rng.standard_normal((mmax, nlat, bc)) + 1j * rng.standard_normal((mmax, nlat, bc))).astype(np.complex64)).to(dev)
    full = ops.legendre_fwd_raw(xf, tabw, lmax, mode=mode)
    m_off, mloc = 9, 11
    part = ops.legendre_fwd_raw(xf[m_off:m_off + mloc].contiguous(), tabw, lmax, m_off, mode=mode)
    mask = torch.from_numpy(tril_mask(lmax, mloc, 0, m_off)).to(dev)[:, :, None]
    assert torch.equal(torch.where(mask, part, 0), torch.where(mask, full[:, m_off:m_off + mloc], 0))
    cfull = torch.where(torch.from_numpy(tril_mask(lmax, mmax)).to(dev)[:, :, None], full, 0)
    yfull = ops.legendre_inv_raw(cfull, tabp, nlat, mode=mode)
    ypart = ops.legendre_inv_raw(cfull[:, m_off:m_off + mloc].contiguous(), tabp, nlat, m_off, mode=mode)
    assert torch.equal(ypart, yfull[m_off:m_off + mloc])


# --------------------------------------------------------------------------- full SHT vs oracle
SHT_CASES = [("equiangular", 33, 64, 16, 17, (2, 3)), ("legendre-gauss", 32, 64, 32, 33, (5,)),
             ("equiangular", 91, 180, 30, 31, (1, 2)), ("legendre-gauss", 240, 480, 240, 241, (3,)),
             ("equiangular", 721, 1440, 240, 241, (2,))]


@pytest.mark.parametrize("grid,nlat,nlon,lmax,mmax,lead", SHT_CASES)
def test_sht_modules_vs_oracle(dev, grid, nlat, nlon, lmax, mmax, lead):
    from makani_amd.sht import RealSHT, InverseRealSHT
    from oracle import sht as osht
    rng = np.random.default_rng(333)
    x = rng.standard_normal((*lead, nlat, nlon)).astype(np.float32)
    f = RealSHT(nlat, nlon, lmax, mmax, grid).to(dev).float()
    fi = InverseRealSHT(nlat, nlon, lmax, mmax, grid).to(dev).float()
    c = f(torch.from_numpy(x).to(dev))
    assert c.shape == (*lead, lmax, mmax) and c.dtype == torch.complex64
    want = osht.RealSHT(nlat, nlon, lmax, mmax, grid, dtype=np.float64)(x)
    assert rel(c.cpu().numpy(), want) < TOL
    # exact zeros where l < m, as the reference's zero table entries give
    cz = c.cpu().numpy()
    assert np.all(cz[..., ~tril_mask(lmax, mmax)] == 0)
    y = fi(c)
    wanty = osht.InverseRealSHT(nlat, nlon, lmax, mmax, grid, dtype=np.float64)(want)
    assert y.shape == x.shape and y.dtype == torch.float32
    assert rel(y.cpu().numpy(), wanty) < TOL
    # projection idempotence (size independent property): sht(isht(sht x)) == sht x
    c2 = f(y)
    assert rel(c2.cpu().numpy(), c.cpu().numpy()) < TOL


def test_sht_autograd_vs_oracle(dev):
    from makani_amd.sht import RealSHT, InverseRealSHT
    from oracle import spectral as osp
    g = torch.Generator().manual_seed(11)
    nlat, nlon, lmax, mmax, grid = 33, 64, 16, 17, "equiangular"
    x = torch.randn(2, 3, nlat, nlon, generator=g)
    f, fi = RealSHT(nlat, nlon, lmax, mmax, grid).to(dev), InverseRealSHT(nlat, nlon, lmax, mmax, grid).to(dev)
    fo, fio = osp.TorchRealSHT(nlat, nlon, lmax, mmax, grid), osp.TorchInverseRealSHT(nlat, nlon, lmax, mmax, grid)
    xd = x.to(dev).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    wgt = torch.randn(2, 3, nlat, nlon, generator=g)
    cw = torch.complex(torch.randn(2, 3, lmax, mmax, generator=g), torch.randn(2, 3, lmax, mmax, generator=g))
    c = f(xd)
    loss = (fi(c) * wgt.to(dev)).sum() + (c * cw.to(dev)).real.sum()
    loss.backward()
    co = fo(xo)
    losso = (fio(co) * wgt).sum() + (co * cw).real.sum()
    losso.backward()
    assert abs(loss.item() - losso.item()) / abs(losso.item()) < 1e-4
    assert rel(xd.grad.cpu().numpy(), xo.grad.numpy()) < TOL


# --------------------------------------------------------------------------- dhconv
DH_CASES = [  # L, M, B, I, O, l_off, m_off
    (16, 17, 2, 6, 5, 0, 0),       # odd O -> float2 path
    (12, 13, 1, 8, 8, 0, 0),
    (40, 41, 3, 24, 40, 0, 0),     # several row tiles
    (10, 21, 2, 16, 12, 20, 0),    # l shard (h-parallel)
    (30, 11, 1, 12, 16, 0, 15),    # m shard (w-parallel): low degrees have no valid mode
    (70, 71, 1, 72, 130, 0, 0),    # > 1 column tile, ragged
    (140, 141, 1, 8, 10, 0, 0),    # > 128 rows (m, b): two row tiles of the bf16x3 engine
    (9, 10, 2, 136, 70, 0, 0),     # > 128 input channels: two wgrad row tiles
]


def _dh_ref(x, w, l_off, m_off):
    # x [L,M,B,I], w [I,O,L] -> y [L,M,B,O], zero where l < m
    y = np.einsum("lmbi,iol->lmbo", x.astype(np.complex128), w.astype(np.complex128))
    mask = tril_mask(x.shape[0], x.shape[1], l_off, m_off)[:, :, None, None]
    return np.where(mask, y, 0), mask


@pytest.mark.parametrize("mode", GEMM_MODES)
@pytest.mark.parametrize("L,M,B,I,O,l_off,m_off", DH_CASES)
def test_dhconv_fwd_bwd(dev, L, M, B, I, O, l_off, m_off, mode):
    from makani_amd import ops
    rng = np.random.default_rng(21)

    def crand(*s):
        return (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)

    x, w, gy = crand(L, M, B, I), crand(I, O, L), crand(L, M, B, O)
    want, mask = _dh_ref(x, w, l_off, m_off)
    wphys = torch.from_numpy(np.ascontiguousarray(w.transpose(2, 0, 1))).to(dev)
    xd = torch.from_numpy(x.reshape(L, M, B * I)).to(dev)
    y = ops.dhconv_fwd_raw(xd, wphys, B, l_off, m_off, mode=mode).cpu().numpy().reshape(L, M, B, O)
    assert rel(np.where(mask, y, 0), want) < TOL
    # dgrad / wgrad against the analytic adjoints (torch convention: conj on the other operand)
    gyd = torch.from_numpy(gy.reshape(L, M, B * O)).to(dev)
    gym = np.where(mask, gy, 0).astype(np.complex128)
    gx = ops.dhconv_dgrad_raw(gyd, wphys, B, l_off, m_off, mode=mode).cpu().numpy().reshape(L, M, B, I)
    want_gx = np.einsum("lmbo,iol->lmbi", gym, np.conj(w).astype(np.complex128))
    assert rel(np.where(mask, gx, 0), want_gx) < TOL
    gw = ops.dhconv_wgrad_raw(xd, gyd, B, l_off, m_off, mode=mode).cpu().numpy()  # [L,I,O]
    want_gw = np.einsum("lmbi,lmbo->lio", np.conj(np.where(mask, x, 0)).astype(np.complex128), gym)
    assert rel(gw, want_gw) < TOL


def test_dhconv_autograd_matches_reference_einsum(dev, golden_dir):
    """The op behind the reference's `_contract_dhconv` golden vector + torch autograd parity."""
    import os
    from makani_amd import ops
    g = np.load(os.path.join(golden_dir, "ref_contractions.npz"))
    x, w, want = g["x"], g["w_dhconv"], g["y_dhconv"]       # x [B,I,X,Y], w [I,O,X]
    B, I, X, Y = x.shape
    O = w.shape[1]
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    wt = torch.from_numpy(w).to(dev).requires_grad_(True)
    xp = ops.spec_pack(xt.reshape(B * I, X, Y))
    yp = ops.dhconv(xp, wt, B)
    y = ops.spec_unpack(yp).reshape(B, O, X, Y)
    mask = tril_mask(X, Y)
    assert rel(np.where(mask, y.detach().cpu().numpy(), 0), np.where(mask, want, 0)) < TOL
    gy = torch.from_numpy(np.where(mask, g["y_diagonal"][:, :O], 0).astype(np.complex64)).to(dev)
    y.backward(gy)
    xo = torch.from_numpy(np.where(mask, x, 0)).requires_grad_(True)
    wo = torch.from_numpy(w).requires_grad_(True)
    yo = torch.einsum("bixy,iox->boxy", xo, wo)
    yo.backward(gy.cpu())
    assert rel(np.where(mask, xt.grad.cpu().numpy(), 0), xo.grad.numpy()) < TOL
    assert rel(wt.grad.cpu().numpy(), wo.grad.numpy()) < TOL


# --------------------------------------------------------------------------- layout
@pytest.mark.parametrize("bc,L,M", [(3, 16, 17), (70, 33, 40), (1, 1, 1)])
def test_pack_unpack(dev, bc, L, M):
    from makani_amd import ops
    rng = np.random.default_rng(2)
    c = (rng.standard_normal((bc, L, M)) + 1j * rng.standard_normal((bc, L, M))).astype(np.complex64)
    ct = torch.from_numpy(c).to(dev)
    p = ops.spec_pack_raw(ct)
    assert np.array_equal(p.cpu().numpy(), c.transpose(1, 2, 0))
    u = ops.spec_unpack_raw(p, 0, 0).cpu().numpy()
    assert np.array_equal(u, np.where(tril_mask(L, M), c, 0))
    u2 = ops.spec_unpack_raw(p, 3, 5).cpu().numpy()
    assert np.array_equal(u2, np.where(tril_mask(L, M, 3, 5), c, 0))


# --------------------------------------------------------------------------- fused pointwise ops
PW_CASES = [(2, 6, 33, 64), (1, 5, 16, 24), (3, 4, 91, 8), (1, 3, 240, 480)]


@pytest.mark.parametrize("B,C,H,W", PW_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_gelu(dev, B, C, H, W, dtype):
    from makani_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).to(dtype)
    b = torch.randn(C, generator=g)
    gy = torch.randn(B, C, H, W, generator=g).to(dtype)
    xd, bd = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert ops.pointwise_supported(xd)
    y = ops.bias_gelu(xd, bd)
    y.backward(gy.to(dev))
    xo, bo = x.double().requires_grad_(True), b.double().requires_grad_(True)
    yo = torch.nn.functional.gelu(xo + bo.view(1, -1, 1, 1))
    yo.backward(gy.double())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and rel(y.detach().float().cpu().numpy(), yo.detach().numpy()) < tol
    assert rel(xd.grad.float().cpu().numpy(), xo.grad.numpy()) < tol
    assert rel(bd.grad.cpu().numpy(), bo.grad.numpy()) < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("B,C,H,W", PW_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fuse", [False, True])
def test_instance_norm(dev, B, C, H, W, dtype, fuse):
    from makani_amd import ops
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, C, H, W, generator=g) * 3 + 5).to(dtype)      # mean >> 0: exercises the variance formula
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    gy = torch.randn(B, C, H, W, generator=g).to(dtype)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.instance_norm(xd, wd, bd, 1e-6, fuse)
    y.backward(gy.to(dev))
    xo, wo, bo = (t.double().requires_grad_(True) for t in (x, w, b))
    yo = torch.nn.functional.instance_norm(xo, weight=wo, bias=bo, eps=1e-6)
    if fuse:
        yo = torch.nn.functional.gelu(yo)
    yo.backward(gy.double())
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and rel(y.detach().float().cpu().numpy(), yo.detach().numpy()) < tol
    assert rel(xd.grad.float().cpu().numpy(), xo.grad.numpy()) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert rel(wd.grad.cpu().numpy(), wo.grad.numpy()) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert rel(bd.grad.cpu().numpy(), bo.grad.numpy()) < (1e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("B,O,I,P", [(1, 128, 128, 4096), (2, 73, 384, 33 * 64), (1, 768, 384, 240 * 480), (3, 5, 7, 24),
                                      (1, 384, 73, 16384 + 8),
                                      (1, 768, 384, 400008), (2, 384, 768, 200008)])   # large-block kernels (256x192 / 192x256)
def test_conv1x1_wgrad(dev, B, O, I, P):
    from makani_amd import _lib, ops
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(B, O, P, generator=g).to(torch.bfloat16)
    x = torch.randn(B, I, P, generator=g).to(torch.bfloat16)
    gyd, xd = gy.to(dev), x.to(dev)
    gw = torch.zeros(O, I, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().mk_conv1x1_wgrad(gyd.data_ptr(), xd.data_ptr(), gw.data_ptr(), B, O, I, P, ops._stream()))
    want = torch.einsum("bop,bip->oi", gy.double(), x.double())
    assert rel(gw.cpu().numpy(), want.numpy()) < 2e-6      # exact bf16 products, fp32 accumulation


# --------------------------------------------------------------------------- fp32 1x1 convolutions on the bf16x3 engine
@pytest.mark.parametrize("B,M,K,P", [(1, 384, 73, 4104), (2, 73, 384, 33 * 64), (1, 768, 384, 240 * 480), (3, 5, 7, 24), (1, 384, 768, 16392),
                                      (1, 200, 130, 2050)])
def test_conv1x1_x3(dev, B, M, K, P):
    """mk_conv1x1_x3: forward, in-place accumulation onto an addend, and the weight gradient of a 1x1 convolution on fp32 fields,
    against float64 (fp32-accurate: six bf16 MFMA products of exact three-way operand splits)."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(5)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    x = torch.randn(B, K, P, generator=g)
    add = torch.randn(B, M, P, generator=g)
    gy = torch.randn(B, M, P, generator=g)
    want = torch.matmul(w.double(), x.double())
    y = ops.conv1x1_x3(w.to(dev), x.to(dev))
    assert rel(y.cpu().numpy(), want.numpy()) < 2e-6
    out = add.to(dev).clone()
    y2 = ops.conv1x1_x3(w.to(dev), x.to(dev), out=out)
    assert y2.data_ptr() == out.data_ptr()
    assert rel(y2.cpu().numpy(), (want + add.double()).numpy()) < 2e-6
    if P % 4 == 0:
        gw = ops.conv1x1_x3_wgrad(gy.to(dev), x.to(dev))
        assert rel(gw.cpu().numpy(), torch.einsum("bop,bip->oi", gy.double(), x.double()).numpy()) < 2e-6


# --------------------------------------------------------------------------- diagonal filter contraction
@pytest.mark.parametrize("B,I,O,L,M", [(2, 6, 5, 7, 8), (5, 3, 9, 16, 17), (1, 32, 32, 30, 31)])
def test_diag_contract_fwd_bwd(dev, B, I, O, L, M):
    """mk_diag_* vs the reference einsum `bixy,ioxy->boxy` (contractions.py:121-127) and its autograd gradients."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(11)

    def crand(*s):
        return torch.complex(torch.randn(*s, generator=g), torch.randn(*s, generator=g))

    x, w, gy = crand(B, I, L, M), crand(I, O, L, M), crand(B, O, L, M)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.diag_contract(xd, wd)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(dev))
    xr, wr = x.to(torch.complex128).requires_grad_(True), w.to(torch.complex128).requires_grad_(True)
    yr = torch.einsum("bixy,ioxy->boxy", xr, wr)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.to(torch.complex128))
    assert rel(y.detach().cpu().numpy(), yr.detach().numpy()) < TOL
    assert rel(gx.cpu().numpy(), gxr.numpy()) < TOL
    assert rel(gw.cpu().numpy(), gwr.numpy()) < TOL


# --------------------------------------------------------------------------- latitude-major Fourier rows (distributed SHT)
@pytest.mark.parametrize("nlat,nlon,mmax,bc", [(9, 480, 33, 5), (7, 1440, 241, 3), (6, 64, 20, 4), (5, 30, 16, 2)])
def test_fft_latitude_major_layout(dev, nlat, nlon, mmax, bc):
    """xf_layout = 1 ([K][M][BC]) holds exactly the numbers of the default [M][K][BC] layout, split / planned / generic kernels."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(bc, nlat, nlon, generator=g).to(dev)
    tw = ops.fft_twiddles(nlon).to(dev)
    s = 2 * math.pi / nlon
    a = ops.rfft_raw(x, tw, mmax, s, s, s)
    b = ops.rfft_raw(x, tw, mmax, s, s, s, kmajor=True)
    assert tuple(b.shape) == (nlat, mmax, bc) and torch.equal(a.permute(1, 0, 2), b)
    y0 = ops.irfft_raw(a, tw, nlon, 1.0, 1.0, 1.0)
    y1 = ops.irfft_raw(b, tw, nlon, 1.0, 1.0, 1.0, kmajor=True)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("grid,nlat,lmax,mmax,bc", [("equiangular", 33, 16, 17, 3), ("legendre-gauss", 240, 240, 241, 6),
                                                     ("equiangular", 65, 70, 40, 66)])
def test_legendre_latitude_major_layout(dev, grid, nlat, lmax, mmax, bc):
    from makani_amd import ops
    rng = np.random.default_rng(9)
    tab = ops.legendre_table(grid, nlat, lmax, mmax, True).to(dev)
    xf = torch.from_numpy((rng.standard_normal((mmax, nlat, bc)) + 1j * rng.standard_normal((mmax, nlat, bc))).astype(np.complex64)).to(dev)
    mask = torch.from_numpy(tril_mask(lmax, mmax)).to(dev)[:, :, None]
    c0 = ops.legendre_fwd_raw(xf, tab, lmax, mode="bf16x3")
    c1 = ops.legendre_fwd_raw(xf.permute(1, 0, 2).contiguous(), tab, lmax, kmajor=True)
    assert torch.equal(torch.where(mask, c0, 0), torch.where(mask, c1, 0))
    cm = torch.where(mask, c0, 0)
    y0 = ops.legendre_inv_raw(cm, tab, nlat, mode="bf16x3")
    y1 = ops.legendre_inv_raw(cm, tab, nlat, kmajor=True)
    assert tuple(y1.shape) == (nlat, mmax, bc) and torch.equal(y0.permute(1, 0, 2), y1)


# --------------------------------------------------------------------------- fused latitude-weighted MSE (bench loss)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weighted_mse(dev, dtype):
    from makani_amd import ops
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 5, 19, 64
    pred = torch.randn(B, C, H, W, generator=g).to(dtype).to(dev).requires_grad_(True)
    tar = torch.randn(B, C, H, W, generator=g).to(dev)
    w = torch.rand(H, generator=g).to(dev)
    loss = ops.weighted_mse(pred, tar, w, 0.25)
    (gp,) = torch.autograd.grad(loss, pred)
    pr = pred.detach().double().requires_grad_(True)
    want = 0.25 * (((pr - tar.double()) ** 2) * w.double().view(1, 1, -1, 1)).sum()
    (gw,) = torch.autograd.grad(want, pr)
    assert abs(loss.item() - want.item()) < 1e-5 * abs(want.item())
    tol = 1e-6 if dtype == torch.float32 else 4e-3      # the gradient is rounded to pred's dtype
    assert rel(gp.float().cpu().numpy(), gw.cpu().numpy()) < tol


@pytest.mark.parametrize("nlat,nlon,mmax,B,C,cpp", [(5, 480, 33, 2, 48, 24), (3, 1440, 241, 1, 96, 48), (4, 480, 241, 3, 24, 24)])
def test_fft_peer_major_layout(dev, nlat, nlon, mmax, B, C, cpp):
    """mk_rfft_pm / mk_irfft_pm: [C/cpp][K][M][B][cpp] holds exactly the numbers of the latitude-major layout."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B * C, nlat, nlon, generator=g).to(dev)
    tw = ops.fft_twiddles(nlon).to(dev)
    s = 2 * math.pi / nlon
    a = ops.rfft_raw(x, tw, mmax, s, s, s, kmajor=True)                         # [K, M, B*C]
    b = ops.rfft_pm_raw(x, tw, mmax, s, s, s, C, cpp)                             # [P, K, M, B*cpp]
    want = a.view(nlat, mmax, B, C // cpp, cpp).permute(3, 0, 1, 2, 4).reshape(C // cpp, nlat, mmax, B * cpp)
    assert torch.equal(b, want)
    y0 = ops.irfft_raw(a, tw, nlon, 1.0, 1.0, 1.0, kmajor=True)
    y1 = ops.irfft_pm_raw(b, tw, nlon, 1.0, 1.0, 1.0, C, cpp)
    assert torch.equal(y0, y1)
    yb = ops.irfft_pm_raw(b, tw, nlon, 1.0, 1.0, 1.0, C, cpp, torch.bfloat16)
    assert torch.equal(yb, ops.irfft_raw(a, tw, nlon, 1.0, 1.0, 1.0, torch.bfloat16, kmajor=True))


@pytest.mark.parametrize("nlat,nlon,mmax,B,C,cpp", [(5, 480, 33, 2, 48, 24), (3, 1440, 241, 1, 96, 48), (7, 480, 241, 1, 20, 0),
                                                     (2, 1440, 200, 2, 9, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_irfft_row_statistics(dev, nlat, nlon, mmax, B, C, cpp, dtype):
    """mk_irfft_sums: the inverse FFT's by-product -- per output row (b, c) the sum and the sum of squares over latitudes and
    longitudes, ON THE VALUES AS STORED (bf16 rows: the rounded ones) -- against float64 sums of the rows it wrote; the rows
    themselves bit-identical to the plain launch.  Peer-major and latitude-major Fourier rows, ragged channel tiles (20, 9),
    truncated modes."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(16)
    tw = ops.fft_twiddles(nlon).to(dev)
    xf = torch.complex(torch.randn(nlat, mmax, B * C, generator=g), torch.randn(nlat, mmax, B * C, generator=g)).to(dev)
    plain = ops.irfft_raw(xf, tw, nlon, 1.0, 1.0, 1.0, dtype, kmajor=True)
    if cpp:
        pm = xf.view(nlat, mmax, B, C // cpp, cpp).permute(3, 0, 1, 2, 4).reshape(C // cpp, nlat, mmax, B * cpp).contiguous()
        x, sums = ops.irfft_sums_raw(pm, tw, nlon, dtype, True, C, cpp)
    else:
        x, sums = ops.irfft_sums_raw(xf, tw, nlon, dtype, True)
    assert torch.equal(x, plain) and sums.shape == (B * C, 2) and sums.dtype == torch.float64
    xd = x.double()
    want = torch.stack([xd.sum(dim=(1, 2)), (xd * xd).sum(dim=(1, 2))], dim=1)
    # fp32 partial sums per 480-real sub-row, fp64 across rows: relative to the sum of |x| resp. x^2
    scale = torch.stack([xd.abs().sum(dim=(1, 2)), (xd * xd).sum(dim=(1, 2))], dim=1)
    assert ((sums - want).abs() / scale).max().item() < 2e-6
    # the autograd op: (x, sums), gradient of x only
    xf2 = xf.clone().requires_grad_(True)
    y, s2 = ops.irfft(xf2, tw, nlon, dtype, True, True)
    assert torch.equal(s2, sums) and not s2.requires_grad
    y.float().square().sum().backward()
    xf3 = xf.clone().requires_grad_(True)
    ops.irfft(xf3, tw, nlon, dtype, True).float().square().sum().backward()
    assert torch.equal(xf2.grad, xf3.grad)


@pytest.mark.parametrize("M,K,P,B,bias,gelu", [(384, 384, 1000, 2, True, True), (768, 73, 520, 1, True, False), (73, 384, 264, 3, False, True),
                                                 (130, 200, 136, 1, True, True)])
def test_conv1x1_x3_bias_gelu_epilogue(dev, M, K, P, B, bias, gelu):
    """mk_conv1x1_x3_bias_act: the fp32 convolution with the bias add and the exact GELU in its epilogue (the inference path of
    `nn.Conv2d(.., 1)` + `nn.GELU()`, layers.py:95-99, 158-206) against the float64 composition."""
    from makani_amd import ops
    g = torch.Generator().manual_seed(12)
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev)
    x = torch.randn(B, K, P, generator=g).to(dev)
    b = torch.randn(M, generator=g).to(dev) if bias else None
    y = ops.conv1x1_x3(w, x, bias=b, gelu=gelu)
    want = torch.einsum("mk,bkp->bmp", w.double(), x.double())
    if bias:
        want = want + b.double().view(1, -1, 1)
    if gelu:
        want = torch.nn.functional.gelu(want)
    assert rel(y.cpu().numpy(), want.cpu().numpy()) < 2e-6


def test_adam_step_matches_torch(dev):
    """mk_adam_step (one streaming pass, makani_amd/optim.py) against torch.optim.Adam over three steps, on a real
    tensor, a complex one and a permuted-contiguous one (the dhconv weight layout), with weight decay."""
    from makani_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(1100, 1001), (96, 96, 120)]
    ps = [torch.randn(*shapes[0], device=dev), torch.randn(*shapes[1], dtype=torch.complex64, device=dev),
          torch.randn(120, 96, 96, dtype=torch.complex64, device=dev).permute(1, 2, 0), torch.randn(7, 5, device=dev)]
    ps = [torch.nn.Parameter(p.clone() if p.is_contiguous() else p) for p in ps]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedAdam(ps, lr=1e-2, weight_decay=0.1)
    assert len(opt._big) == 3 and len(opt._small) == 1
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=0.1)
    for _ in range(3):
        for p, r in zip(ps, ref):
            g = torch.randn_like(r)
            r.grad = g.clone()
            p.grad = g.clone() if p.is_contiguous() else g.permute(2, 0, 1).contiguous().permute(1, 2, 0)
        opt.step()
        ropt.step()
    for p, r in zip(ps, ref):
        a, b = torch.view_as_real(p.detach()) if p.is_complex() else p.detach(), torch.view_as_real(r.detach()) if r.is_complex() else r.detach()
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-6


def test_adam_overlapped_with_backward_is_the_same_update(dev):
    """FusedAdam(overlap_backward=1) launches the update of a large tensor from its post-accumulate hook on a side stream
    (under the rest of backward) and joins it in step(): parameters and moments must equal, bit for bit, those of the
    optimizer that does everything in step() -- over several steps, with a tensor whose gradient arrives early in backward
    (so later backward kernels overlap its update), a complex one, zero_grad in both modes and a skipped gradient."""
    from makani_amd.optim import FusedAdam
    torch.manual_seed(5)

    def make():
        torch.manual_seed(11)
        a = torch.nn.Parameter(torch.randn(1100, 1001, device=dev))
        b = torch.nn.Parameter(torch.randn(96, 96, 120, dtype=torch.complex64, device=dev))
        c = torch.nn.Parameter(torch.randn(1001, 1300, device=dev))          # used first in forward: its gradient comes last
        d = torch.nn.Parameter(torch.randn(64, device=dev))                  # small: torch's fused Adam
        return [a, b, c, d]

    def loss_fn(ps, x, z, use_b=True):
        a, b, c, d = ps
        h = torch.tanh(x @ c[:, :1100])                      # [32, 1100]
        y = (h @ a) * 1e-2                                   # [32, 1001]
        s = (torch.view_as_real(b * z) ** 2).mean() if use_b else 0.0
        return (y ** 2).mean() + s + (d ** 2).sum() * 1e-3

    xs = [torch.randn(32, 1001, device=dev) for _ in range(4)]
    zs = [torch.randn(96, 96, 120, dtype=torch.complex64, device=dev) for _ in range(4)]
    runs = []
    for overlap in (0, 1):
        ps = make()
        opt = FusedAdam(ps, lr=1e-2, weight_decay=0.05, overlap_backward=overlap)
        assert len(opt._big) == 3 and len(opt._small) == 1
        for i in range(4):
            opt.zero_grad(set_to_none=(i % 2 == 0))
            loss = loss_fn(ps, xs[i], zs[i], use_b=(i != 2))       # step 2: the complex tensor gets no gradient at all
            loss.backward()
            if overlap:
                assert any(st["done"] for st in opt._big)
            opt.step()
            assert not any(st["done"] for st in opt._big)
        torch.cuda.synchronize()
        runs.append(([p.detach().clone() for p in ps], [st["m"].clone() for st in opt._big], [st["v"].clone() for st in opt._big],
                     [st["step"] for st in opt._big]))
    for pa, pb in zip(runs[0][0], runs[1][0]):
        assert torch.equal(torch.view_as_real(pa) if pa.is_complex() else pa, torch.view_as_real(pb) if pb.is_complex() else pb)
    for k in (1, 2):
        for ta, tb in zip(runs[0][k], runs[1][k]):
            assert torch.equal(ta, tb)
    assert runs[0][3] == runs[1][3]


def test_conv_kernels_reject_misaligned_and_odd_sizes(dev):
    """The bf16 1x1-conv kernels read 16-byte vectors: odd pixel counts and offset views whose base is not 16-byte
    aligned must come back as an error from the C ABI (never as a device fault), and aligned offset views must work.
    (Round-1 note: the one GPU memory fault on record, gpurun_out/tun.log, happened while torch's TunableOp was trying
    vendor GEMM solutions on the first transposed 73-channel shape -- 146-byte rows; the in-tree kernels were not running.)"""
    from makani_amd import ops, _lib
    lib = _lib.load()
    torch.manual_seed(0)
    O, I, P = 40, 24, 264
    gy = torch.randn(1, O, P + 8, device=dev).bfloat16()
    x = torch.randn(1, I, P + 8, device=dev).bfloat16()
    gw = torch.zeros(O, I, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # odd pixel count
    assert lib.mk_conv1x1_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), 1, O, I, P + 3, st) != 0
    # base pointer off by one element (2 bytes)
    assert lib.mk_conv1x1_wgrad(gy.data_ptr() + 2, x.data_ptr(), gw.data_ptr(), 1, O, I, P, st) != 0
    assert lib.mk_pce_gemm(x.data_ptr() + 2, x.data_ptr(), gy.data_ptr(), None, None, None, None, 0, 1, O, I, P, st) != 0
    # an aligned offset view (8 elements = 16 bytes in) is fine and exact
    gyv = gy.view(-1)[8:8 + O * P].view(1, O, P)
    xv = x.view(-1)[8:8 + I * P].view(1, I, P)
    assert gyv.data_ptr() % 16 == 0 and gyv.is_contiguous()
    got = ops.conv1x1_wgrad_raw(gyv, xv)
    want = gyv[0].double() @ xv[0].double().t()
    assert rel(got.cpu().numpy(), want.cpu().numpy()) < 1e-5
    w = (torch.randn(O, I, device=dev) / I ** 0.5)
    y = ops.pce_gemm(xv, ops.pce_pack(w), O)
    assert rel(y.float().cpu().numpy(), (w.bfloat16().double() @ xv[0].double()).unsqueeze(0).cpu().numpy()) < 3e-3


def test_geometric_l2_loss_fused_pass(dev):
    """The absolute squared geometric L2 loss with uniform channel weights is one HIP pass (mk_wmse_*); it must agree with
    the oracle and with the general torch path (non-uniform weights), values and gradients."""
    from makani_amd.losses import GeometricLpLoss
    from oracle import losses as ol
    g = torch.Generator().manual_seed(12)
    B, C, H, W = 2, 6, 48, 96
    prd, tar = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    q = ol.quad_weight("legendre-gauss", (H, W), (H, W), (0, 0), normalize=True)
    loss = GeometricLpLoss((H, W), (H, W), (0, 0), p=2, absolute=True, squared=True, quadrature_rule="legendre-gauss").to(dev)
    for chw, uniform in ((torch.full((1, C), 1.0 / C), 1.0 / C), (torch.rand(1, C, generator=g), None)):
        loss.uniform_chw = uniform             # what LossHandler knows on the host; None = general torch path
        pd = prd.to(dev).requires_grad_(True)
        out = loss(pd, tar.to(dev), chw.to(dev))
        out.backward()
        assert (type(out.grad_fn).__name__ == "_WeightedMSEBackward") == (uniform is not None)
        want = ol.geometric_lp_loss(prd.numpy(), tar.numpy(), chw.numpy(), q, p=2, absolute=True, squared=True)
        assert abs(float(out) - want) < 5e-6 * abs(want)
        gwant = 2.0 * (prd - tar).double() * torch.from_numpy(q) * chw.double().view(1, C, 1, 1)
        assert rel(pd.grad.cpu().numpy(), gwant.numpy()) < 2e-6
