"""Autograd operators over the C ABI (include/makani_amd.h).

Tensors cross the boundary as raw device pointers + sizes; torch is only the
allocator and the stream owner.  Every launch goes to torch's *current* HIP
stream, so the ops are capturable in a HIP graph exactly like the reference's
step is captured in ``makani/utils/trainer.py:84-152``.

Private layouts (see the header): ``xf`` = complex64 ``[M, K, BC]``, spectrum =
complex64 ``[L, M, BC]`` (channels last), dhconv weight = complex64 ``[L, I, O]``.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib

GRIDS = {"equiangular": 0, "legendre-gauss": 1}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("makani_amd: the spectral ops run only on a HIP device (got a CPU tensor); "
                               "there is no CPU fallback")


# ----------------------------------------------------------------------------
# host-side precompute (float64 in C, no GPU needed)
# ----------------------------------------------------------------------------
def quadrature(grid, nlat):
    """(colatitudes ascending from the north pole, quadrature weights) as float64 numpy."""
    if grid not in GRIDS:
        raise ValueError(f"Unknown quadrature mode {grid}")
    lib = _lib.load()
    theta = np.empty(nlat, dtype=np.float64)
    w = np.empty(nlat, dtype=np.float64)
    _lib.check(lib.mk_quadrature(GRIDS[grid], nlat, theta.ctypes.data, w.ctypes.data), "mk_quadrature")
    return theta, w


def legendre_kpad(nlat):
    return _lib.load().mk_legendre_kpad(nlat)


def legendre_table(grid, nlat, lmax, mmax, with_quad_weights):
    """fp32 table [mmax, lmax, kpad] (zero padded along k) as a CPU torch tensor."""
    if grid not in GRIDS:
        raise ValueError(f"Unknown quadrature mode {grid}")
    lib = _lib.load()
    kp = lib.mk_legendre_kpad(nlat)
    out = torch.empty(mmax, lmax, kp, dtype=torch.float32)
    _lib.check(lib.mk_legendre_table(GRIDS[grid], nlat, lmax, mmax, int(bool(with_quad_weights)), out.data_ptr()),
               "mk_legendre_table")
    return out


def fft_twiddles(nlon):
    lib = _lib.load()
    out = torch.empty(lib.mk_fft_twiddle_len(nlon), dtype=torch.float32)
    _lib.check(lib.mk_fft_twiddles(nlon, out.data_ptr()), "mk_fft_twiddles")
    return out


# ----------------------------------------------------------------------------
# raw (non-differentiable) launches
# ----------------------------------------------------------------------------
def rfft_raw(x, twiddles, mmax, s0, sm, sh, kmajor=False):
    """x real [BC, K, N] (fp32 or bf16) -> xf complex64 [mmax, K, BC], or [K, mmax, BC] with ``kmajor``."""
    _need_cuda(x, twiddles)
    assert x.dim() == 3 and x.is_contiguous()
    bc, k, n = x.shape
    if x.dtype == torch.float32:
        dt = 0
    elif x.dtype == torch.bfloat16:
        dt = 1
    else:
        raise TypeError(f"mk_rfft: unsupported dtype {x.dtype}")
    xf = torch.empty((k, mmax, bc) if kmajor else (mmax, k, bc), dtype=torch.complex64, device=x.device)
    _lib.check(_lib.load().mk_rfft_ex(x.data_ptr(), dt, xf.data_ptr(), twiddles.data_ptr(), bc, k, n, mmax,
                                      s0, sm, sh, int(bool(kmajor)), _stream()), "mk_rfft")
    return xf


def irfft_bf16_rows(nlon, mmax):
    """bf16 output rows are built for the production lengths (the split kernels) only."""
    return nlon in (480, 1440) and mmax <= 241


def irfft_sums_supported(nlon, mmax):
    """The inverse FFT can deliver the row statistics of its output (``mk_irfft_sums``: the split kernels)."""
    return nlon in (480, 1440) and mmax <= 241 and os.environ.get("MK_FFT_LEGACY", "0") != "1" \
        and os.environ.get("MK_IRFFT_SUMS", "1") != "0"


def irfft_sums_raw(xf, twiddles, nlon, out_dtype, kmajor=False, chans=0, cpp=0):
    """``irfft`` (scales 1, 1, 1) of plain (``[M, K, BC]`` / ``[K, M, BC]``) or peer-major (``cpp`` > 0:
    ``[chans / cpp, K, M, B * cpp]``) Fourier rows -> (x ``[BC, K, nlon]``, fp64 ``[BC, 2]`` sums and sums of squares of its rows)."""
    _need_cuda(xf, twiddles)
    assert xf.is_contiguous() and xf.dtype == torch.complex64 and out_dtype in (torch.float32, torch.bfloat16)
    if cpp:
        p, k, m, bcp = xf.shape
        bc = (bcp // cpp) * chans
    elif kmajor:
        k, m, bc = xf.shape
    else:
        m, k, bc = xf.shape
    x = torch.empty(bc, k, nlon, dtype=out_dtype, device=xf.device)
    sums = torch.zeros(bc, 2, dtype=torch.float64, device=xf.device)
    _lib.check(_lib.load().mk_irfft_sums(xf.data_ptr(), x.data_ptr(), 1 if out_dtype == torch.bfloat16 else 0, twiddles.data_ptr(),
                                         bc, k, nlon, m, 1.0, 1.0, 1.0, int(bool(kmajor)), int(chans), int(cpp), sums.data_ptr(),
                                         _stream()), "mk_irfft_sums")
    return x, sums


def irfft_raw(xf, twiddles, nlon, s0, sm, sh, out_dtype=torch.float32, kmajor=False):
    """xf complex64 [M, K, BC] ([K, M, BC] with ``kmajor``) -> x [BC, K, nlon] in fp32, or bf16 where the kernel
    fuses the cast."""
    _need_cuda(xf, twiddles)
    assert xf.dim() == 3 and xf.is_contiguous() and xf.dtype == torch.complex64
    if kmajor:
        k, m, bc = xf.shape
    else:
        m, k, bc = xf.shape
    fused = out_dtype == torch.bfloat16 and irfft_bf16_rows(nlon, m)
    x = torch.empty(bc, k, nlon, dtype=torch.bfloat16 if fused else torch.float32, device=xf.device)
    _lib.check(_lib.load().mk_irfft_ex(xf.data_ptr(), x.data_ptr(), 1 if fused else 0, twiddles.data_ptr(), bc, k, nlon, m,
                                       s0, sm, sh, int(bool(kmajor)), _stream()), "mk_irfft")
    return x if x.dtype == out_dtype else x.to(out_dtype)


def fft_pm_supported(nlon, mmax, chans, chans_per_peer):
    """Peer-major Fourier rows (mk_rfft_pm / mk_irfft_pm): split kernels, even channel blocks that are multiples of 24."""
    return (irfft_bf16_rows(nlon, mmax) and chans_per_peer > 0 and chans % chans_per_peer == 0 and chans_per_peer % 24 == 0
            and os.environ.get("MK_FFT_LEGACY") != "1")


def rfft_pm_raw(x, twiddles, mmax, s0, sm, sh, chans, chans_per_peer):
    """x real [B*chans, K, N] -> xf complex64 [chans / cpp, K, mmax, B * cpp] (peer-major, see the header)."""
    _need_cuda(x, twiddles)
    assert x.dim() == 3 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    bc, k, n = x.shape
    xf = torch.empty(chans // chans_per_peer, k, mmax, (bc // chans) * chans_per_peer, dtype=torch.complex64, device=x.device)
    _lib.check(_lib.load().mk_rfft_pm(x.data_ptr(), 0 if x.dtype == torch.float32 else 1, xf.data_ptr(), twiddles.data_ptr(),
                                      bc, k, n, mmax, s0, sm, sh, chans, chans_per_peer, _stream()), "mk_rfft_pm")
    return xf


def irfft_pm_raw(xf, twiddles, nlon, s0, sm, sh, chans, chans_per_peer, out_dtype=torch.float32):
    """xf complex64 [chans / cpp, K, M, B * cpp] -> x [B*chans, K, nlon] (fp32 or bf16 rows)."""
    _need_cuda(xf, twiddles)
    assert xf.dim() == 4 and xf.is_contiguous() and xf.dtype == torch.complex64
    p, k, m, bcp = xf.shape
    bc = (bcp // chans_per_peer) * chans
    fused = out_dtype == torch.bfloat16
    x = torch.empty(bc, k, nlon, dtype=torch.bfloat16 if fused else torch.float32, device=xf.device)
    _lib.check(_lib.load().mk_irfft_pm(xf.data_ptr(), x.data_ptr(), 1 if fused else 0, twiddles.data_ptr(), bc, k, nlon, m,
                                       s0, sm, sh, chans, chans_per_peer, _stream()), "mk_irfft_pm")
    return x if x.dtype == out_dtype else x.to(out_dtype)


# Arithmetic of the spectral GEMMs (Legendre, dhconv): "bf16x3" = exact three-way bf16 split of every fp32
# operand, six bf16 MFMA products, fp32 accumulation (fp32-accurate, csrc/gemm_x3.hip); "f32" = fp32 MFMA
# (csrc/gemm.hip).  Both meet the 1e-5 parity budget; bf16x3 is the faster one on gfx950.
SPECTRAL_GEMM = os.environ.get("MK_SPECTRAL_GEMM", "bf16x3")


def _gemm_mode(mode):
    mode = mode or SPECTRAL_GEMM
    if mode not in ("bf16x3", "f32"):
        raise ValueError(f"unknown spectral GEMM mode {mode!r} (bf16x3 | f32)")
    return mode


def legendre_x3_image(table, nlat, inverse):
    """Pre-split tile image of a device fp32 Legendre table (built once, cached on the tensor object).

    The image is built on whatever stream needs it first; the cache entry carries the event recorded behind the
    build, and a hit from another stream waits for it (two micro-batch streams start cold together, pipeline.py)."""
    _need_cuda(table)
    cache = table.__dict__.setdefault("_mk_x3", {})
    key = (int(nlat), int(bool(inverse)))
    entry = cache.get(key)
    if entry is None:
        lib = _lib.load()
        mg, lmax, kp = table.shape
        assert kp == legendre_kpad(nlat) and table.is_contiguous() and table.dtype == torch.float32
        nbytes = lib.mk_legendre_x3_bytes(nlat, lmax, mg, key[1])
        img = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
        _lib.check(lib.mk_legendre_x3_split(table.data_ptr(), img.data_ptr(), nlat, lmax, mg, key[1], _stream()),
                   "mk_legendre_x3_split")
        ev = torch.cuda.Event()
        ev.record()
        cache[key] = entry = [img, ev]
        return img
    if entry[1] is not None:
        if torch.cuda.is_current_stream_capturing():
            # inside a stream capture the event may neither be queried nor waited for (it belongs to uncaptured work:
            # hipErrorStreamCaptureIsolation).  Nothing to do: the build was issued before the capture began, either on this
            # stream (ordered before every replay) or on one the caller has joined before capturing, as for any other input
            return entry[0]
        elif entry[1].query():
            entry[1] = None                     # built and finished: nothing to order any more
        else:
            torch.cuda.current_stream().wait_event(entry[1])
    return entry[0]


def legendre_fwd_raw(xf, table, lmax, m_off=0, mode=None, kmajor=False):
    """xf [Mloc, K, BC] ([K, Mloc, BC] with ``kmajor``: bf16x3 kernels only) -> spectrum [lmax, Mloc, BC]; rows l < m
    are left unwritten."""
    _need_cuda(xf, table)
    assert xf.is_contiguous() and xf.dtype == torch.complex64 and table.dtype == torch.float32
    if kmajor:
        k, mloc, bc = xf.shape
        mode = "bf16x3"
    else:
        mloc, k, bc = xf.shape
    mg, lt, kp = table.shape
    assert lt == lmax and kp == legendre_kpad(k), "Legendre table does not match the operand"
    c = torch.empty(lmax, mloc, bc, dtype=torch.complex64, device=xf.device)
    if _gemm_mode(mode) == "bf16x3":
        img = legendre_x3_image(table, k, 0)
        _lib.check(_lib.load().mk_legendre_fwd_x3_ex(xf.data_ptr(), img.data_ptr(), c.data_ptr(), bc, k, lmax, mloc,
                                                     m_off, mg, int(bool(kmajor)), _stream()), "mk_legendre_fwd_x3")
    else:
        _lib.check(_lib.load().mk_legendre_fwd(xf.data_ptr(), table.data_ptr(), c.data_ptr(), bc, k, lmax, mloc,
                                               m_off, mg, _stream()), "mk_legendre_fwd")
    return c


def legendre_inv_raw(c, table, nlat, m_off=0, mode=None, kmajor=False):
    """spectrum [L, Mloc, BC] -> xf [Mloc, nlat, BC] ([nlat, Mloc, BC] with ``kmajor``: bf16x3 kernels only)."""
    _need_cuda(c, table)
    assert c.is_contiguous() and c.dtype == torch.complex64 and table.dtype == torch.float32
    lmax, mloc, bc = c.shape
    mg, lt, kp = table.shape
    assert lt == lmax and kp == legendre_kpad(nlat), "Legendre table does not match the operand"
    xf = torch.empty((nlat, mloc, bc) if kmajor else (mloc, nlat, bc), dtype=torch.complex64, device=c.device)
    if kmajor:
        mode = "bf16x3"
    if _gemm_mode(mode) == "bf16x3":
        img = legendre_x3_image(table, nlat, 1)
        _lib.check(_lib.load().mk_legendre_inv_x3_ex(c.data_ptr(), img.data_ptr(), xf.data_ptr(), bc, nlat, lmax, mloc,
                                                     m_off, mg, int(bool(kmajor)), _stream()), "mk_legendre_inv_x3")
    else:
        _lib.check(_lib.load().mk_legendre_inv(c.data_ptr(), table.data_ptr(), xf.data_ptr(), bc, nlat, lmax, mloc,
                                               m_off, mg, _stream()), "mk_legendre_inv")
    return xf


def spec_pack_raw(c_std):
    """[BC, L, M] complex64 -> [L, M, BC]."""
    _need_cuda(c_std)
    assert c_std.dim() == 3 and c_std.is_contiguous() and c_std.dtype == torch.complex64
    bc, l, m = c_std.shape
    out = torch.empty(l, m, bc, dtype=torch.complex64, device=c_std.device)
    _lib.check(_lib.load().mk_spec_pack(c_std.data_ptr(), out.data_ptr(), bc, l, m, _stream()), "mk_spec_pack")
    return out


def spec_unpack_raw(c_prv, l_off=0, m_off=0):
    """[L, M, BC] complex64 -> [BC, L, M], exact zeros where l_off + l < m_off + m."""
    _need_cuda(c_prv)
    assert c_prv.dim() == 3 and c_prv.is_contiguous() and c_prv.dtype == torch.complex64
    l, m, bc = c_prv.shape
    out = torch.empty(bc, l, m, dtype=torch.complex64, device=c_prv.device)
    _lib.check(_lib.load().mk_spec_unpack(c_prv.data_ptr(), out.data_ptr(), bc, l, m, l_off, m_off, _stream()),
               "mk_spec_unpack")
    return out


def _w_phys(w):
    """Physical [L, I, O] contiguous view/copy of a dhconv weight of logical shape [I, O, L]."""
    wp = w.permute(2, 0, 1)
    return wp if wp.is_contiguous() else wp.contiguous()


def _dh_fn(name, mode, cin, cout):
    """bf16x3 kernels need even channel counts; odd ones take the fp32 MFMA kernels."""
    if _gemm_mode(mode) == "bf16x3" and cin % 2 == 0 and cout % 2 == 0:
        name += "_x3"
    return getattr(_lib.load(), name), name


def dhconv_fwd_raw(x, w_phys, batch, l_off=0, m_off=0, mode=None):
    _need_cuda(x, w_phys)
    assert x.is_contiguous() and x.dtype == torch.complex64 and w_phys.is_contiguous() and w_phys.dtype == torch.complex64
    lloc, mloc, bc = x.shape
    l2, cin, cout = w_phys.shape
    assert l2 == lloc and bc == batch * cin, "dhconv operand shapes do not match"
    y = torch.empty(lloc, mloc, batch * cout, dtype=torch.complex64, device=x.device)
    fn, name = _dh_fn("mk_dhconv_fwd", mode, cin, cout)
    _lib.check(fn(x.data_ptr(), w_phys.data_ptr(), y.data_ptr(), lloc, mloc, batch, cin, cout, l_off, m_off, _stream()), name)
    return y


def dhconv_dgrad_raw(gy, w_phys, batch, l_off=0, m_off=0, mode=None):
    _need_cuda(gy, w_phys)
    assert gy.is_contiguous() and gy.dtype == torch.complex64 and w_phys.is_contiguous()
    lloc, mloc, bo = gy.shape
    l2, cin, cout = w_phys.shape
    assert l2 == lloc and bo == batch * cout
    gx = torch.empty(lloc, mloc, batch * cin, dtype=torch.complex64, device=gy.device)
    fn, name = _dh_fn("mk_dhconv_dgrad", mode, cin, cout)
    _lib.check(fn(gy.data_ptr(), w_phys.data_ptr(), gx.data_ptr(), lloc, mloc, batch, cin, cout, l_off, m_off, _stream()), name)
    return gx


def dhconv_wgrad_raw(x, gy, batch, l_off=0, m_off=0, mode=None):
    _need_cuda(x, gy)
    assert x.is_contiguous() and gy.is_contiguous() and x.dtype == torch.complex64 and gy.dtype == torch.complex64
    lloc, mloc, bi = x.shape
    cin, cout = bi // batch, gy.shape[2] // batch
    gw = torch.empty(lloc, cin, cout, dtype=torch.complex64, device=x.device)
    fn, name = _dh_fn("mk_dhconv_wgrad", mode, cin, cout)
    _lib.check(fn(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), lloc, mloc, batch, cin, cout, l_off, m_off, _stream()), name)
    return gw


def _diag_launch(name, a, b, out, batch, cin, cout, p):
    _lib.check(getattr(_lib.load(), name)(a.data_ptr(), b.data_ptr(), out.data_ptr(), batch, cin, cout, p, _stream()), name)
    return out


class _DiagContract(torch.autograd.Function):
    """y[b,o,l,m] = sum_i x[b,i,l,m] w[i,o,l,m] on the public layout (mk_diag_*)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        b, i, l, m = x.shape
        o = w.shape[1]
        y = torch.empty(b, o, l, m, dtype=torch.complex64, device=x.device)
        return _diag_launch("mk_diag_fwd", x, w, y, b, i, o, l * m)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        b, i, l, m = x.shape
        o = w.shape[1]
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _diag_launch("mk_diag_dgrad", gy, w, torch.empty_like(x), b, i, o, l * m)
        if ctx.needs_input_grad[1]:
            gw = _diag_launch("mk_diag_wgrad", x, gy, torch.empty_like(w), b, i, o, l * m)
        return gx, gw


def diag_contract(x, w):
    """``einsum("bixy,ioxy->boxy")`` for complex64 x [B,I,L,M], w [I,O,L,M] on the HIP streaming kernels."""
    _need_cuda(x, w)
    if x.dtype != torch.complex64 or w.dtype != torch.complex64:
        raise TypeError("diag_contract expects complex64 operands")
    if x.dim() != 4 or w.dim() != 4 or x.shape[1] != w.shape[0] or tuple(x.shape[2:]) != tuple(w.shape[2:]):
        raise ValueError(f"diag_contract: incompatible shapes {tuple(x.shape)} and {tuple(w.shape)}")
    return _DiagContract.apply(x.contiguous(), w.contiguous())


# ----------------------------------------------------------------------------
# 1x1 convolutions on fp32 fields: the bf16x3 engine of the spectral GEMMs (fp32-accurate, no vendor GEMM)
# ----------------------------------------------------------------------------
def _pad4(w):
    """[M, K] fp32 -> contiguous [M, K4] with K4 = K rounded up to 4 (zero columns): 16-byte aligned rows for the row stager."""
    k4 = (w.shape[1] + 3) // 4 * 4
    if k4 == w.shape[1] and w.is_contiguous():
        return w
    out = w.new_zeros(w.shape[0], k4)
    out[:, :w.shape[1]].copy_(w)
    return out


def conv1x1_x3_supported(x3):
    """fp32 ``[B, K, P]`` contiguous field on the GPU with an even pixel count."""
    return x3.is_cuda and x3.dtype == torch.float32 and x3.dim() == 3 and x3.is_contiguous() and x3.shape[2] % 2 == 0


def conv1x1_x3(w, x3, out=None, bias=None, gelu=False):
    """y[b] = w @ x3[b] (``out`` given: ``out[b] += w @ x3[b]`` in place): w fp32 ``[M, K]``, x3 fp32 ``[B, K, P]`` (``mk_conv1x1_x3``).
    ``bias`` / ``gelu`` (not with ``out``): ``y = act(w @ x3 + bias)`` in the kernel's epilogue (``mk_conv1x1_x3_bias_act``)."""
    _need_cuda(w, x3)
    assert w.dtype == torch.float32 and w.dim() == 2 and conv1x1_x3_supported(x3) and w.shape[1] == x3.shape[1]
    b, k, p = x3.shape
    m = w.shape[0]
    a = _pad4(w)
    if bias is not None or gelu:
        assert out is None
        bf = None if bias is None else bias.detach().float().contiguous()
        assert bf is None or bf.numel() == m
        y = torch.empty(b, m, p, dtype=torch.float32, device=x3.device)
        _lib.check(_lib.load().mk_conv1x1_x3_bias_act(a.data_ptr(), a.stride(0), x3.data_ptr(), p, y.data_ptr(), p, m, k, p, b, k * p,
                                                      m * p, None if bf is None else bf.data_ptr(), int(bool(gelu)), _stream()),
                   "mk_conv1x1_x3_bias_act")
        return y
    if out is not None:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (b, m, p)
    y = out if out is not None else torch.empty(b, m, p, dtype=torch.float32, device=x3.device)
    _lib.check(_lib.load().mk_conv1x1_x3(a.data_ptr(), a.stride(0), x3.data_ptr(), p, y.data_ptr(), p, m, k, p, b, 0, k * p,
                                         m * p, 1 if out is not None else 0, _stream()), "mk_conv1x1_x3")
    return y


def conv1x1_x3_wgrad(gy, x3):
    """gW[o][i] = sum_{b,p} gy[b][o][p] x3[b][i][p] for fp32 fields (P a multiple of 4): fp32 ``[O, I]``."""
    _need_cuda(gy, x3)
    assert gy.dtype == torch.float32 and x3.dtype == torch.float32 and gy.is_contiguous() and x3.is_contiguous()
    b, o, p = gy.shape
    i = x3.shape[1]
    assert p % 4 == 0 and x3.shape[0] == b and x3.shape[2] == p
    gw = torch.zeros(o, i, dtype=torch.float32, device=gy.device)
    _lib.check(_lib.load().mk_conv1x1_x3(gy.data_ptr(), p, x3.data_ptr(), p, gw.data_ptr(), i, o, p, i, b, o * p, i * p, 0, 2,
                                         _stream()), "mk_conv1x1_x3")
    return gw


# ----------------------------------------------------------------------------
# per-step arena of the pointwise stack: every packed weight image in ONE launch, every weight-gradient buffer in ONE fill
# ----------------------------------------------------------------------------
_ACTIVE_ARENA = None


class EngineArena:
    """What the 1x1 convolutions of one net need per step besides their GEMMs, batched: the MFMA fragment images of every
    weight in both orientations (``mk_pce_pack_batch``: one launch instead of one ``mk_pce_pack`` per GEMM) and the zeroed
    fp32 buffers the weight-gradient kernels accumulate into (one fill instead of one per layer).

    ``with arena.scope():`` around a forward pass refreshes the images from the CURRENT weights (always: no invalidation
    protocol to get wrong) and makes ``pce_pack`` / ``conv1x1_wgrad_raw`` find them; the autograd nodes take what their
    backward pass needs (transposed image, gradient buffer) while the scope is open.  Outside a scope -- a layer called on its
    own, a checkpoint recomputation -- every call packs / allocates for itself as before."""

    def __init__(self, weights):
        # detached: only addresses and shapes are kept.  (A `weight.view(out, in)` of a parameter carries a grad_fn that holds
        # the parameter's AccumulateGrad node; keeping it across steps pins that node to the stream of the first step and a
        # later capture on another stream dies in capture_end -- the round-1 crash)
        self.weights = [w.detach() for w in weights if w.is_cuda and w.dim() == 2 and w.stride(1) == 1
                        and w.dtype in (torch.float32, torch.bfloat16)]
        lib = _lib.load()
        rows, self.slots, off, goff = [], {}, 0, 0
        self.gslots = {}
        self.n_fwd = self.total_fwd = 0
        for transpose in (False, True):            # the forward images first: inference (grad mode off) packs only those
            for w in self.weights:
                m, k = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
                nbytes = lib.mk_pce_image_bytes(m, k)
                if nbytes <= 0:
                    continue
                lay = (ctypes.c_longlong * 3)()
                _lib.check(lib.mk_pce_pack_layout(m, k, lay), "mk_pce_pack_layout")
                rows.append([w.data_ptr(), 0 if w.dtype == torch.float32 else 1, int(transpose), m, k, w.stride(0), lay[0], lay[1],
                             lay[2], off])
                self.slots[self._key(w, transpose)] = (off * 2, nbytes)
                off += nbytes // 2
            if not transpose:
                self.n_fwd, self.total_fwd = len(rows), off
        for w in self.weights:
            self.gslots[self._key(w, False)] = (goff, w.shape[0], w.shape[1])
            goff += (w.numel() + 3) // 4 * 4           # 16-byte aligned buffers
        self.total, self.gtotal = off, goff
        self.n = len(rows)
        dev = self.weights[0].device if self.weights else None
        self.desc = torch.tensor(rows + [[0] * 9 + [off]], dtype=torch.int64).to(dev) if rows else None
        self.images = torch.empty(2 * off, dtype=torch.uint8, device=dev) if rows else None
        self.gbuf = None
        self._prev = None

    @staticmethod
    def _key(w, transpose):
        return (w.data_ptr(), tuple(w.shape), w.stride(0), w.dtype, bool(transpose))

    def refresh(self, with_grad_buffers):
        n, total = (self.n, self.total) if with_grad_buffers else (self.n_fwd, self.total_fwd)
        self._packed_all = bool(with_grad_buffers)
        if n:
            _lib.check(_lib.load().mk_pce_pack_batch(self.desc.data_ptr(), n, self.images.data_ptr(), total, _stream()),
                       "mk_pce_pack_batch")
        # a fresh buffer per step: the views handed out become the parameters' gradients and live as long as those do
        self.gbuf = torch.zeros(self.gtotal, dtype=torch.float32, device=self.images.device) if (with_grad_buffers and self.gtotal) else None

    def image(self, w, transpose):
        if transpose and not self._packed_all:       # grad mode was off at scope entry: the transposed images are stale
            return None
        slot = self.slots.get(self._key(w, transpose))
        return None if slot is None else self.images[slot[0]:slot[0] + slot[1]]

    def grad_buffer(self, w):
        """Zeroed fp32 ``[out, in]`` buffer for the gradient of ``w`` -- once per step (a second request gets None: the
        caller then allocates, e.g. a weight used twice in one forward pass)."""
        slot = self.gslots.get(self._key(w, False))
        if slot is None or self.gbuf is None or slot[0] in self._taken:
            return None
        self._taken.add(slot[0])
        return self.gbuf[slot[0]:slot[0] + slot[1] * slot[2]].view(slot[1], slot[2])

    def scope(self):
        return _ArenaScope(self)


class _ArenaScope:
    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global _ACTIVE_ARENA
        self.arena._prev, _ACTIVE_ARENA = _ACTIVE_ARENA, self.arena
        self.arena._taken = set()
        self.arena.refresh(torch.is_grad_enabled())
        return self.arena

    def __exit__(self, *exc):
        global _ACTIVE_ARENA
        _ACTIVE_ARENA = self.arena._prev
        return False


def arena_image(w, transpose=False):
    """The packed image of ``w`` from the open arena scope, or None."""
    return None if _ACTIVE_ARENA is None else _ACTIVE_ARENA.image(w, transpose)


def arena_grad_buffer(w):
    return None if _ACTIVE_ARENA is None else _ACTIVE_ARENA.grad_buffer(w)


def conv1x1_wgrad_raw(gy, x3, x_gelu=False, out=None):
    """gW[o][i] = sum_{b,p} gy[b][o][p] act(x3[b][i][p]): bf16 [B,O,P], [B,I,P] -> fp32 [O,I] (HIP bf16 MFMA kernel);
    ``x_gelu``: act = exact GELU rounded to bf16, applied while x3 is staged (x3 is then the kept pre-activation of an MLP;
    O <= 384).  ``out``: accumulate into this zeroed buffer instead of a fresh one."""
    _need_cuda(gy, x3)
    assert gy.is_contiguous() and x3.is_contiguous() and gy.dtype == torch.bfloat16 and x3.dtype == torch.bfloat16
    b, o, p = gy.shape
    i = x3.shape[1]
    if out is not None:      # a zeroed fp32 [O, I] buffer (EngineArena.grad_buffer)
        assert out.dtype == torch.float32 and tuple(out.shape) == (o, i) and out.is_contiguous()
    gw = out if out is not None else torch.zeros(o, i, dtype=torch.float32, device=x3.device)
    _lib.check(_lib.load().mk_conv1x1_wgrad_act(gy.data_ptr(), x3.data_ptr(), gw.data_ptr(), b, o, i, p, int(bool(x_gelu)),
                                                _stream()), "mk_conv1x1_wgrad_act")
    return gw


# ----------------------------------------------------------------------------
# pixel-column engine (csrc/pce.hip): 1x1 convolutions with fused epilogues
# ----------------------------------------------------------------------------
def pce_supported(m, k):
    """One GEMM ``[m, k] @ [k, P]`` the engine is built for: K <= 768, M <= 1536."""
    return _lib.load().mk_pce_image_bytes(int(m), int(k)) > 0


def pce_supported_train(out_channels, in_channels):
    """A 1x1 convolution the engine can run in BOTH directions: forward ``[out, in]`` and the data gradient, the
    transposed GEMM ``[in, out]`` (K' = out <= 768, M' = in <= 1536).  Layers that pass only the forward check
    (e.g. 768 -> 1536, the fc1 of ``sfno_dhealy_73ch_edim768``) must take the fallback path as a whole: the
    autograd node packs the transposed image in backward."""
    return pce_supported(out_channels, in_channels) and pce_supported(in_channels, out_channels)


def pce_pack(w, transpose=False):
    """MFMA fragment image of A = w (``[M, K]``) or A = w^T (``w`` is ``[K, M]``); fp32 or bf16 weights.  Inside an
    ``EngineArena`` scope that holds ``w`` the image comes from the arena's batched launch."""
    _need_cuda(w)
    img = arena_image(w, transpose)
    if img is not None:
        return img
    assert w.dim() == 2 and w.stride(1) == 1 and w.dtype in (torch.float32, torch.bfloat16)
    m, k = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    lib = _lib.load()
    nbytes = lib.mk_pce_image_bytes(m, k)
    if nbytes <= 0:
        raise ValueError(f"pce_pack: unsupported GEMM shape M={m}, K={k}")
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.mk_pce_pack(w.data_ptr(), 0 if w.dtype == torch.float32 else 1, int(bool(transpose)), m, k, w.stride(0),
                               img.data_ptr(), _stream()), "mk_pce_pack")
    return img


def pce_mlp_supported(m, hd, k1):
    """Shapes of the fused conv -> GELU -> conv node (csrc/pce_mlp.hip): K1 <= 384, Hd <= 768, M <= 384."""
    return _lib.load().mk_pce_mlp_image_bytes(int(m), int(hd), int(k1)) > 0


def pce_mlp_pack(a1, a1_transposed, a2, a2_transposed):
    """Weight stream image of the fused node: ``A1 [Hd, K1]`` = ``a1`` (or ``a1^T`` when ``a1_transposed``: ``a1`` is
    ``[K1, Hd]``), ``A2 [M, Hd]`` = ``a2`` (or ``a2^T``: ``a2`` is ``[Hd, M]``).  Returns ``(image, M, Hd, K1)``."""
    _need_cuda(a1, a2)
    assert a1.dim() == 2 and a2.dim() == 2 and a1.stride(1) == 1 and a2.stride(1) == 1 and a1.dtype == a2.dtype
    assert a1.dtype in (torch.float32, torch.bfloat16)
    hd, k1 = (a1.shape[1], a1.shape[0]) if a1_transposed else (a1.shape[0], a1.shape[1])
    m, hd2 = (a2.shape[1], a2.shape[0]) if a2_transposed else (a2.shape[0], a2.shape[1])
    assert hd == hd2, "the two matrices do not share the hidden dimension"
    lib = _lib.load()
    nbytes = lib.mk_pce_mlp_image_bytes(m, hd, k1)
    if nbytes <= 0:
        raise ValueError(f"pce_mlp_pack: unsupported shape M={m}, Hd={hd}, K1={k1}")
    img = torch.empty(nbytes, dtype=torch.uint8, device=a1.device)
    _lib.check(lib.mk_pce_mlp_pack(a1.data_ptr(), int(bool(a1_transposed)), a1.stride(0), a2.data_ptr(), int(bool(a2_transposed)),
                                   a2.stride(0), 0 if a1.dtype == torch.float32 else 1, m, hd, k1, img.data_ptr(), _stream()),
               "mk_pce_mlp_pack")
    return img, m, hd, k1


def pce_mlp(x3, packed, mode, b1=None, b2=None, pre=None, want_row_sums=False, want_mid_sums=False):
    """The fused node on bf16 ``[B, K1, P]`` fields (``mk_pce_mlp``).  ``mode`` 0: returns ``(y, pre[, sums])`` with
    ``pre = A1 x + b1`` and ``y = A2 gelu(pre) + b2``; ``mode`` 1 (``x3`` = output gradient, ``pre`` = the kept pre-activation):
    returns ``(gx, gpre[, gpre_sums])`` with ``gpre = (A1 x) * gelu'(pre)``, ``gx = A2 gpre`` and the fp64 ``[B * Hd]`` pixel sums
    of ``gpre``."""
    img, m, hd, k1 = packed
    _need_cuda(x3, img)
    assert x3.dim() == 3 and x3.is_contiguous() and x3.dtype == torch.bfloat16 and x3.shape[1] == k1
    b, _, p = x3.shape
    if mode == 1:
        assert pre is not None and pre.is_contiguous() and pre.dtype == torch.bfloat16 and tuple(pre.shape) == (b, hd, p)
    bf1 = None if b1 is None else b1.detach().float().contiguous()
    bf2 = None if b2 is None else b2.detach().float().contiguous()
    assert (bf1 is None or bf1.numel() == hd) and (bf2 is None or bf2.numel() == m)
    y = torch.empty(b, m, p, dtype=torch.bfloat16, device=x3.device)
    mid = torch.empty(b, hd, p, dtype=torch.bfloat16, device=x3.device)
    sums = torch.empty(b * m, 2, dtype=torch.float64, device=x3.device) if want_row_sums else None
    msum = torch.empty(b * hd, dtype=torch.float64, device=x3.device) if want_mid_sums else None
    _lib.check(_lib.load().mk_pce_mlp(x3.data_ptr(), img.data_ptr(), y.data_ptr(), mid.data_ptr(),
                                      None if pre is None else pre.data_ptr(), None if bf1 is None else bf1.data_ptr(),
                                      None if bf2 is None else bf2.data_ptr(), None if sums is None else sums.data_ptr(),
                                      None if msum is None else msum.data_ptr(), int(mode), b, m, hd, k1, p, _stream()),
               "mk_pce_mlp")
    return (y, mid) + ((sums,) if want_row_sums else ()) + ((msum,) if want_mid_sums else ())


_ZERO_BIAS = {}


def _zero_bias(device):
    z = _ZERO_BIAS.get(device)
    if z is None:
        z = _ZERO_BIAS[device] = torch.zeros(1024, dtype=torch.float32, device=device)
    return z


def pce_gemm(x3, wimg, m, bias=None, addend=None, aux_in=None, want_pre=False, gelu=False, want_row_sums=False,
             addend_affine=None):
    """y[b] = epi(A @ x3[b]) on bf16 ``[B, K, P]`` fields (see ``mk_pce_gemm_ex``).  Returns ``y``, followed by ``pre``
    (the bf16 pre-activation ``A x + bias``) when ``want_pre`` and by the fp64 ``[B * M, 2]`` row sums (sum, sum of squares
    over the pixels of ``y``) when ``want_row_sums``.  ``addend_affine`` (fp32 ``[B * M, 2]``) lets the addend enter as
    ``a * addend + b`` per row (``instance_norm_coeffs``)."""
    _need_cuda(x3, wimg)
    assert x3.dim() == 3 and x3.is_contiguous() and x3.dtype == torch.bfloat16
    b, k, p = x3.shape
    for t in (addend, aux_in):
        if t is not None:
            assert t.is_contiguous() and t.dtype == torch.bfloat16 and tuple(t.shape) == (b, m, p)
    bf = _zero_bias(x3.device)   # the epilogue loads a bias unconditionally: zeros for layers without one
    if bias is not None:      # fp32 [m]
        bf = bias.detach().float().contiguous()
        assert bf.numel() == m
    y = torch.empty(b, m, p, dtype=torch.bfloat16, device=x3.device)
    pre = torch.empty_like(y) if want_pre else None
    sums = torch.empty(b * m, 2, dtype=torch.float64, device=x3.device) if want_row_sums else None
    if addend_affine is not None:
        assert (addend is not None and addend_affine.dtype == torch.float32 and addend_affine.is_contiguous()
                and addend_affine.numel() == 2 * b * m)
    _lib.check(_lib.load().mk_pce_gemm_ex(x3.data_ptr(), wimg.data_ptr(), y.data_ptr(), bf.data_ptr(),
                                          None if addend is None else addend.data_ptr(),
                                          None if addend_affine is None else addend_affine.data_ptr(),
                                          None if aux_in is None else aux_in.data_ptr(),
                                          None if pre is None else pre.data_ptr(), int(bool(gelu)),
                                          None if sums is None else sums.data_ptr(), b, m, k, p, _stream()),
               "mk_pce_gemm_ex")
    out = (y,) + ((pre,) if want_pre else ()) + ((sums,) if want_row_sums else ())
    return out if len(out) > 1 else y


# ----------------------------------------------------------------------------
# differentiable operators (all linear in the data: backward = adjoint launch)
# ----------------------------------------------------------------------------
class _RFFT(torch.autograd.Function):
    """x [BC, K, N] -> xf [mmax, K, BC] = 2 pi rfft(x, norm="forward")[..., :mmax] (K1)."""

    @staticmethod
    def forward(ctx, x, twiddles, mmax, kmajor=False):
        ctx.save_for_backward(twiddles)
        ctx.nlon = x.shape[-1]
        ctx.in_dtype = x.dtype
        ctx.kmajor = kmajor
        s = 2.0 * math.pi / ctx.nlon
        return rfft_raw(x, twiddles, mmax, s, s, s, kmajor)

    @staticmethod
    def backward(ctx, gxf):
        (tw,) = ctx.saved_tensors
        n = ctx.nlon
        gx = irfft_raw(gxf.contiguous(), tw, n, 2.0 * math.pi / n, math.pi / n, 2.0 * math.pi / n, ctx.in_dtype, ctx.kmajor)
        return gx, None, None, None


class _IRFFT(torch.autograd.Function):
    """xf [M, K, BC] -> x [BC, K, nlon] = irfft(xf, n=nlon, norm="forward") (K4)."""

    @staticmethod
    def forward(ctx, xf, twiddles, nlon, out_dtype, kmajor=False, want_sums=False):
        ctx.save_for_backward(twiddles)
        ctx.mmax = xf.shape[1] if kmajor else xf.shape[0]
        ctx.kmajor = kmajor
        if want_sums:            # (x, row statistics): the sums are data for the norm's kernel, not a differentiable output
            x, sums = irfft_sums_raw(xf, twiddles, nlon, out_dtype, kmajor)
            ctx.mark_non_differentiable(sums)
            return x, sums
        return irfft_raw(xf, twiddles, nlon, 1.0, 1.0, 1.0, out_dtype, kmajor)

    @staticmethod
    def backward(ctx, gx, *_):
        (tw,) = ctx.saved_tensors
        if gx.dtype not in (torch.float32, torch.bfloat16):
            gx = gx.float()
        return rfft_raw(gx.contiguous(), tw, ctx.mmax, 1.0, 2.0, 1.0, ctx.kmajor), None, None, None, None, None


class _RFFTpm(torch.autograd.Function):
    """rfft into peer-major Fourier rows; adjoint = irfft from them."""

    @staticmethod
    def forward(ctx, x, twiddles, mmax, chans, cpp):
        ctx.save_for_backward(twiddles)
        ctx.cfg = (x.shape[-1], x.dtype, chans, cpp)
        s = 2.0 * math.pi / x.shape[-1]
        return rfft_pm_raw(x, twiddles, mmax, s, s, s, chans, cpp)

    @staticmethod
    def backward(ctx, gxf):
        (tw,) = ctx.saved_tensors
        n, dt, chans, cpp = ctx.cfg
        gx = irfft_pm_raw(gxf.contiguous(), tw, n, 2.0 * math.pi / n, math.pi / n, 2.0 * math.pi / n, chans, cpp, dt)
        return gx, None, None, None, None


class _IRFFTpm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xf, twiddles, nlon, out_dtype, chans, cpp, want_sums=False):
        ctx.save_for_backward(twiddles)
        ctx.cfg = (xf.shape[2], chans, cpp)
        if want_sums:
            x, sums = irfft_sums_raw(xf, twiddles, nlon, out_dtype, True, chans, cpp)
            ctx.mark_non_differentiable(sums)
            return x, sums
        return irfft_pm_raw(xf, twiddles, nlon, 1.0, 1.0, 1.0, chans, cpp, out_dtype)

    @staticmethod
    def backward(ctx, gx, *_):
        (tw,) = ctx.saved_tensors
        mmax, chans, cpp = ctx.cfg
        if gx.dtype not in (torch.float32, torch.bfloat16):
            gx = gx.float()
        return rfft_pm_raw(gx.contiguous(), tw, mmax, 1.0, 2.0, 1.0, chans, cpp), None, None, None, None, None, None


class _LegendreFwd(torch.autograd.Function):
    """xf [Mloc, K, BC] -> c [L, Mloc, BC] with table[m_off + m] (K2)."""

    @staticmethod
    def forward(ctx, xf, table, lmax, m_off, kmajor=False):
        ctx.table = table   # constant buffer; the python object carries the cached bf16x3 images
        ctx.nlat, ctx.m_off, ctx.kmajor = xf.shape[0] if kmajor else xf.shape[1], m_off, kmajor
        return legendre_fwd_raw(xf, table, lmax, m_off, kmajor=kmajor)

    @staticmethod
    def backward(ctx, gc):
        return legendre_inv_raw(gc.contiguous(), ctx.table, ctx.nlat, ctx.m_off, kmajor=ctx.kmajor), None, None, None, None


class _LegendreInv(torch.autograd.Function):
    """c [L, Mloc, BC] -> xf [Mloc, K, BC] with table[m_off + m] (K3)."""

    @staticmethod
    def forward(ctx, c, table, nlat, m_off, kmajor=False):
        ctx.table = table
        ctx.lmax, ctx.m_off, ctx.kmajor = c.shape[0], m_off, kmajor
        return legendre_inv_raw(c, table, nlat, m_off, kmajor=kmajor)

    @staticmethod
    def backward(ctx, gxf):
        return legendre_fwd_raw(gxf.contiguous(), ctx.table, ctx.lmax, ctx.m_off, kmajor=ctx.kmajor), None, None, None, None


class _SpecPack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c_std, l_off, m_off):
        ctx.offs = (l_off, m_off)
        return spec_pack_raw(c_std)

    @staticmethod
    def backward(ctx, g):
        return spec_unpack_raw(g.contiguous(), *ctx.offs), None, None


class _SpecUnpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c_prv, l_off, m_off):
        return spec_unpack_raw(c_prv, l_off, m_off)

    @staticmethod
    def backward(ctx, g):
        return spec_pack_raw(g.contiguous()), None, None


class _Dhconv(torch.autograd.Function):
    """y[l, m, b, o] = sum_i x[l, m, b, i] w[i, o, l] on the private layout (K5)."""

    @staticmethod
    def forward(ctx, x, w, batch, l_off, m_off):
        wp = _w_phys(w)
        ctx.save_for_backward(x, wp)
        ctx.args = (batch, l_off, m_off)
        return dhconv_fwd_raw(x, wp, batch, l_off, m_off)

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        batch, l_off, m_off = ctx.args
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = dhconv_dgrad_raw(gy, wp, batch, l_off, m_off)
        if ctx.needs_input_grad[1]:
            # physical [L, I, O] -> logical [I, O, L] view with the parameter's own strides
            gw = dhconv_wgrad_raw(x, gy, batch, l_off, m_off).permute(1, 2, 0)
        return gx, gw, None, None, None


def rfft(x, twiddles, mmax, kmajor=False):
    return _RFFT.apply(x, twiddles, mmax, kmajor)


def irfft(xf, twiddles, nlon, out_dtype=torch.float32, kmajor=False, want_sums=False):
    """``want_sums``: returns (x, fp64 ``[BC, 2]`` row sums / sums of squares of x) -- see ``mk_irfft_sums``."""
    return _IRFFT.apply(xf, twiddles, nlon, out_dtype, kmajor, want_sums)


def rfft_pm(x, twiddles, mmax, chans, cpp):
    return _RFFTpm.apply(x, twiddles, mmax, chans, cpp)


def irfft_pm(xf, twiddles, nlon, out_dtype, chans, cpp, want_sums=False):
    return _IRFFTpm.apply(xf, twiddles, nlon, out_dtype, chans, cpp, want_sums)


def legendre_fwd(xf, table, lmax, m_off=0, kmajor=False):
    return _LegendreFwd.apply(xf, table, lmax, m_off, kmajor)


def legendre_inv(c, table, nlat, m_off=0, kmajor=False):
    return _LegendreInv.apply(c, table, nlat, m_off, kmajor)


def spec_pack(c_std, l_off=0, m_off=0):
    return _SpecPack.apply(c_std, l_off, m_off)


def spec_unpack(c_prv, l_off=0, m_off=0):
    return _SpecUnpack.apply(c_prv, l_off, m_off)


def dhconv(x, w, batch, l_off=0, m_off=0):
    return _Dhconv.apply(x, w, batch, l_off, m_off)


# ----------------------------------------------------------------------------
# fused pointwise ops of the FNO block (bias + GELU, instance norm [+ GELU])
# ----------------------------------------------------------------------------
def _pw_dtype(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError(f"makani_amd pointwise ops: unsupported dtype {t.dtype}")


def pointwise_supported(x):
    """NCHW, contiguous, on the GPU, fp32/bf16, H*W a multiple of 8, B*C <= 65535."""
    return (x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
            and (x.shape[2] * x.shape[3]) % 8 == 0 and x.shape[0] * x.shape[1] <= 65535)


class _BiasGelu(torch.autograd.Function):
    """y = gelu(x + bias[c]) on [B, C, H, W] (bias may be None)."""

    @staticmethod
    def forward(ctx, x, bias):
        _need_cuda(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        bf = None if bias is None else bias.detach().float().contiguous()
        _lib.check(_lib.load().mk_bias_gelu_fwd(x.data_ptr(), 0 if bf is None else bf.data_ptr(), y.data_ptr(),
                                                _pw_dtype(x), B * C, C, H * W, _stream()), "mk_bias_gelu_fwd")
        ctx.save_for_backward(x, bf if bf is not None else x.new_empty(0))
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, bf = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gb = torch.zeros(C, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        _lib.check(_lib.load().mk_bias_gelu_bwd(x.data_ptr(), bf.data_ptr() if ctx.has_bias else 0, gy.data_ptr(),
                                                gx.data_ptr(), 0 if gb is None else gb.data_ptr(), _pw_dtype(x),
                                                B * C, C, H * W, _stream()), "mk_bias_gelu_bwd")
        return gx, (gb.to(ctx.bias_dtype) if gb is not None else None)


class _InstanceNorm(torch.autograd.Function):
    """Affine instance norm over (H, W) with optional fused GELU; statistics in fp32/fp64.

    ``group`` / ``count``: the rows are sharded over the ranks of ``group`` (spatial model parallelism) -- the local
    row sums are all-reduced between the two kernel phases and ``count`` is the global number of elements per row.
    The weight / bias gradients returned are the LOCAL contributions (shared-weight reduction sums them later).
    """

    @staticmethod
    def forward(ctx, x, weight, bias, eps, fuse_gelu, group, count, sums=None):
        _need_cuda(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        wf = None if weight is None else weight.detach().float().contiguous()
        bf = None if bias is None else bias.detach().float().contiguous()
        stats = torch.empty(B * C, 2, dtype=torch.float32, device=x.device)
        if sums is not None:     # the producer of x delivered the local row sums (engine epilogue): statistics pass done
            assert sums.dtype == torch.float64 and sums.is_contiguous() and sums.numel() == 2 * B * C
        ws = sums if sums is not None else torch.empty(B * C, 2, dtype=torch.float64, device=x.device)
        lib = _lib.load()
        cnt = H * W if group is None else int(count)

        def run(phase):
            _lib.check(lib.mk_instnorm_fwd_ex(x.data_ptr(), 0 if wf is None else wf.data_ptr(),
                                              0 if bf is None else bf.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                              ws.data_ptr(), _pw_dtype(x), B * C, C, H * W, cnt, float(eps),
                                              int(fuse_gelu), phase, _stream()), "mk_instnorm_fwd_ex")
        if group is None:
            run(0 if sums is None else 2)
        else:
            if sums is None:
                run(1)
            torch.distributed.all_reduce(ws, group=group)
            run(2)
        empty = x.new_empty(0, dtype=torch.float32)
        ctx.save_for_backward(x, stats, wf if wf is not None else empty, bf if bf is not None else empty)
        ctx.cfg = (weight is not None, bias is not None, bool(fuse_gelu),
                   None if weight is None else weight.dtype, None if bias is None else bias.dtype, group, cnt)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, wf, bf = ctx.saved_tensors
        has_w, has_b, fuse, wdt, bdt, group, cnt = ctx.cfg
        gx, gw, gb = instance_norm_backward(x, gy, stats, wf if has_w else None, bf if has_b else None, fuse, group, cnt,
                                            grad_dtype=wdt if (has_w and has_b and wdt == bdt) else None)
        return gx, gw.to(wdt) if has_w else None, gb.to(bdt) if has_b else None, None, None, None, None, None


def instance_norm_backward(x, gy, stats, wf, bf, fuse_gelu=False, group=None, count=None, grad_dtype=None):
    """Backward of the (optionally GELU-fused) instance norm from its saved ``stats`` [B * C, 2] = (mean, rstd):
    returns (gx, local weight-gradient sums [C], local bias-gradient sums [C]) -- the sums in fp64, or, with ``grad_dtype``,
    cast to it by ONE copy into a ``[2, C]`` buffer whose rows are the two (contiguous) results."""
    B, C, H, W = x.shape
    gy = gy.contiguous()
    gx = torch.empty_like(x)
    ws = torch.empty(B * C, 2, dtype=torch.float64, device=x.device)
    lib = _lib.load()
    cnt = H * W if group is None else int(count)

    def run(phase):
        _lib.check(lib.mk_instnorm_bwd_ex(x.data_ptr(), gy.data_ptr(), stats.data_ptr(),
                                          wf.data_ptr() if wf is not None else 0, bf.data_ptr() if bf is not None else 0,
                                          gx.data_ptr(), ws.data_ptr(), _pw_dtype(x), B * C, C, H * W, cnt, int(fuse_gelu),
                                          phase, _stream()), "mk_instnorm_bwd_ex")
    if group is None and B == 1 and grad_dtype == torch.float32 and wf is not None and bf is not None:
        # one sample: the kernel writes the affine parameters' gradients itself (no copy / cast launch behind it)
        out = torch.empty(2, C, dtype=torch.float32, device=x.device)
        _lib.check(lib.mk_instnorm_bwd_wb(x.data_ptr(), gy.data_ptr(), stats.data_ptr(), wf.data_ptr(), bf.data_ptr(),
                                          gx.data_ptr(), ws.data_ptr(), out.data_ptr(), _pw_dtype(x), C, H * W, int(fuse_gelu),
                                          _stream()), "mk_instnorm_bwd_wb")
        return gx, out[0], out[1]
    if group is None:
        run(0)
        local = ws
    else:
        run(1)
        local = ws.clone()
        torch.distributed.all_reduce(ws, group=group)
        run(2)
    sums = local.view(B, C, 2)
    sums = sums[0] if B == 1 else sums.sum(0)
    if grad_dtype is not None:
        out = torch.empty(2, C, dtype=grad_dtype, device=x.device)
        out.copy_(sums.t())
        return gx, out[1], out[0]
    return gx, sums[:, 1], sums[:, 0]


def instance_norm_coeffs(sums, weight, bias, rows, channels, count, eps):
    """Row sums fp64 [rows, 2] (global when sharded) -> (stats, affine), both fp32 [rows, 2]: (mean, rstd) and the affine
    map (a, b) with norm(x) = a * x + b (``mk_instnorm_coeffs``)."""
    _need_cuda(sums)
    assert sums.dtype == torch.float64 and sums.is_contiguous() and sums.numel() == 2 * rows
    stats = torch.empty(rows, 2, dtype=torch.float32, device=sums.device)
    affine = torch.empty(rows, 2, dtype=torch.float32, device=sums.device)
    _lib.check(_lib.load().mk_instnorm_coeffs(sums.data_ptr(), 0 if weight is None else weight.data_ptr(),
                                              0 if bias is None else bias.data_ptr(), stats.data_ptr(), affine.data_ptr(),
                                              rows, channels, int(count), float(eps), _stream()), "mk_instnorm_coeffs")
    return stats, affine


class _WeightedMSE(torch.autograd.Function):
    """scale * sum wrow[h] (pred - tar)^2 over [B, C, H, W]; one streaming pass each way (mk_wmse_*)."""

    @staticmethod
    def forward(ctx, pred, tar, wrow, scale):
        _need_cuda(pred, tar, wrow)
        B, C, H, W = pred.shape
        loss = torch.empty(1, dtype=torch.float64, device=pred.device)
        _lib.check(_lib.load().mk_wmse_fwd(pred.data_ptr(), _pw_dtype(pred), tar.data_ptr(), wrow.data_ptr(), loss.data_ptr(),
                                           B * C * H, H, W, float(scale), _stream()), "mk_wmse_fwd")
        ctx.save_for_backward(pred, tar, wrow)
        ctx.scale = float(scale)
        return loss.float().squeeze(0)

    @staticmethod
    def backward(ctx, g):
        pred, tar, wrow = ctx.saved_tensors
        B, C, H, W = pred.shape
        gp = torch.empty_like(pred)
        g32 = g.detach().float().reshape(1).contiguous()
        _lib.check(_lib.load().mk_wmse_bwd(pred.data_ptr(), _pw_dtype(pred), tar.data_ptr(), wrow.data_ptr(), g32.data_ptr(),
                                           gp.data_ptr(), B * C * H, H, W, ctx.scale, _stream()), "mk_wmse_bwd")
        return gp, None, None, None


def weighted_mse(pred, tar, wrow, scale=1.0):
    """``scale * (((pred - tar) ** 2) * wrow[None, None, :, None]).sum()`` for pred [B, C, H, W] (fp32 / bf16, contiguous),
    tar fp32 of the same shape, wrow fp32 [H]; gradient w.r.t. pred only."""
    if not (pred.is_cuda and pred.is_contiguous() and tar.is_contiguous() and tar.dtype == torch.float32
            and pred.dtype in (torch.float32, torch.bfloat16) and pred.shape[-1] % 8 == 0):
        w = wrow.view(1, 1, -1, 1)
        return scale * (((pred.float() - tar) ** 2) * w).sum()
    return _WeightedMSE.apply(pred, tar, wrow.float().contiguous(), scale)


def row_sums(t3):
    """Sum over the last axis of a contiguous [B, C, P] fp32 / bf16 field -> fp32 [B * C] (mk_instnorm_fwd_ex, phase 1)."""
    _need_cuda(t3)
    assert t3.dim() == 3 and t3.is_contiguous()
    b, c, p = t3.shape
    if os.environ.get("MK_ROWSUM") == "torch":
        return torch.sum(t3, dim=-1, dtype=torch.float32).view(-1)
    ws = torch.empty(b * c, 2, dtype=torch.float64, device=t3.device)
    _lib.check(_lib.load().mk_instnorm_fwd_ex(t3.data_ptr(), 0, 0, 0, 0, ws.data_ptr(), _pw_dtype(t3), b * c, c, p, p, 0.0, 0,
                                              1, _stream()), "mk_instnorm_fwd_ex")
    return ws[:, 0].float()


def bias_gelu(x, bias):
    return _BiasGelu.apply(x, bias)


def instance_norm(x, weight, bias, eps=1e-5, fuse_gelu=False, group=None, count=None, row_sums=None):
    """``row_sums``: fp64 ``[B * C, 2]`` LOCAL (sum, sum of squares) of every row of ``x`` when its producer already has
    them (``pce_gemm(..., want_row_sums=True)``); the statistics pass over ``x`` is then skipped."""
    return _InstanceNorm.apply(x, weight, bias, eps, fuse_gelu, group, count, row_sums)
