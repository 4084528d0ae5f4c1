"""RealSHT / InverseRealSHT on MI355X -- the transform seam of the SFNO hot path.

Drop-in for ``torch_harmonics.RealSHT`` / ``InverseRealSHT`` as the reference
constructs them at ``makani/models/networks/sfnonet.py:536-539`` and calls them at
``makani/models/common/spectral_convolution.py:131-141`` / ``sfnonet.py:596-598``:
same constructor arguments, the attributes ``nlat, nlon, lmax, mmax, grid`` that
``SpectralConv`` reads, tables as non-persistent buffers (so reference
checkpoints load with ``strict=True``), arbitrary leading dims, autograd.

The arithmetic is two HIP launches per direction through the C ABI: the batched
longitudinal real FFT (``mk_rfft`` / ``mk_irfft``) and the Legendre contraction on
fp32 MFMA (``mk_legendre_fwd`` / ``mk_legendre_inv``); ``forward_packed`` /
``inverse_packed`` expose the channels-last spectrum ``[L, M, BC]`` that the fused
``SpectralConv`` path consumes without a layout round trip.
"""
import torch
import torch.nn as nn

from . import ops


class _SHTBase(nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__()
        if norm != "ortho" or not csphase:
            raise NotImplementedError("only norm='ortho', csphase=True (what Makani uses) are built")
        if grid not in ops.GRIDS:
            raise ValueError(f"Unknown quadrature mode {grid}")
        if nlon % 2:
            raise ValueError("nlon must be even")
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.norm, self.csphase = norm, csphase
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        if self.mmax > nlon // 2 + 1:
            raise ValueError("mmax exceeds nlon // 2 + 1")
        self.register_buffer("twiddles", ops.fft_twiddles(nlon), persistent=False)

    def _flatten(self, x, inner):
        lead = x.shape[:-2]
        bc = 1
        for s in lead:
            bc *= s
        return lead, x.reshape(bc, *inner)

    def extra_repr(self):
        return f"nlat={self.nlat}, nlon={self.nlon}, lmax={self.lmax}, mmax={self.mmax}, grid={self.grid}"


class RealSHT(_SHTBase):
    """real [..., nlat, nlon] -> complex64 [..., lmax, mmax]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        # weights[m, l, k] = P_l^m(cos theta_k) * w_k, zero padded along k
        self.register_buffer("weights", ops.legendre_table(grid, nlat, self.lmax, self.mmax, True), persistent=False)

    def forward_packed(self, x3):
        """x3 [BC, nlat, nlon] (fp32 or bf16, contiguous) -> spectrum [lmax, mmax, BC]."""
        km = ops.SPECTRAL_GEMM == "bf16x3"   # latitude-major Fourier rows [K, M, BC]: the FFT writes them 10 % faster
        xf = ops.rfft(x3, self.twiddles, self.mmax, km)
        return ops.legendre_fwd(xf, self.weights, self.lmax, 0, km)

    def forward(self, x):
        if x.shape[-2] != self.nlat or x.shape[-1] != self.nlon:
            raise ValueError(f"expected [..., {self.nlat}, {self.nlon}], got {tuple(x.shape)}")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        lead, x3 = self._flatten(x.contiguous(), (self.nlat, self.nlon))
        c = ops.spec_unpack(self.forward_packed(x3), 0, 0)
        return c.reshape(*lead, self.lmax, self.mmax)


class InverseRealSHT(_SHTBase):
    """complex64 [..., lmax, mmax] -> real fp32 [..., nlat, nlon]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self.register_buffer("pct", ops.legendre_table(grid, nlat, self.lmax, self.mmax, False), persistent=False)

    def inverse_packed(self, c, out_dtype=torch.float32, want_row_sums=False):
        """spectrum [lmax, mmax, BC] -> x [BC, nlat, nlon] (fp32, or bf16 rows straight from the FFT kernel); with
        ``want_row_sums`` (and a length the split kernels serve) also the fp64 ``[BC, 2]`` sums / sums of squares of the rows of x
        -- the statistics of the instance norm that reads x next -- else None in their place."""
        km = ops.SPECTRAL_GEMM == "bf16x3"
        xf = ops.legendre_inv(c, self.pct, self.nlat, 0, km)
        if want_row_sums:
            if ops.irfft_sums_supported(self.nlon, self.mmax) and out_dtype in (torch.float32, torch.bfloat16):
                return ops.irfft(xf, self.twiddles, self.nlon, out_dtype, km, True)
            return ops.irfft(xf, self.twiddles, self.nlon, out_dtype, km), None
        return ops.irfft(xf, self.twiddles, self.nlon, out_dtype, km)

    def forward(self, x):
        if x.shape[-2] != self.lmax or x.shape[-1] != self.mmax:
            raise ValueError(f"expected [..., {self.lmax}, {self.mmax}], got {tuple(x.shape)}")
        if x.dtype != torch.complex64:
            x = x.to(torch.complex64)
        lead, c3 = self._flatten(x.contiguous(), (self.lmax, self.mmax))
        y = self.inverse_packed(ops.spec_pack(c3, 0, 0))
        return y.reshape(*lead, self.nlat, self.nlon)
