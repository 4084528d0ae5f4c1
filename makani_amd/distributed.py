"""Spatially distributed SHT: latitude sharded over ``h``, longitude over ``w``.

Drop-in for ``torch_harmonics.distributed.{DistributedRealSHT, DistributedInverseRealSHT,
distributed_transpose_polar, distributed_transpose_azimuth}`` as the reference selects
them at ``makani/models/networks/sfnonet.py:528-533`` (same choreography as the in-tree
``makani/mpu/layers.py:38-169``): the axis being transformed is made local by an
all-to-all that splits the channels instead, the local stage runs (HIP FFT / Legendre
MFMA kernels), and a second all-to-all restores channel-local / mode-sharded data.

MI355X notes: each exchange is ONE ``all_to_all_single`` on a packed contiguous buffer
(uneven shards via split sizes) -- on an 8-GPU xGMI island that is one direct
link per peer -- instead of the reference's list-of-tensors all-to-all.  The
Fourier/spectral intermediates stay in the private channels-last layouts
(``[M, K, B, C]`` / ``[L, M, B, C]``), so the channel split is a split of the
contiguous axis.
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from . import comm, ops
from .sht import _SHTBase


def compute_split_shapes(size, num_chunks):
    """Shard sizes along one axis (modulus ``compute_split_shapes`` as used at mpu/layers.py:64-67)."""
    if num_chunks == 1:
        return [size]
    chunk = (size + num_chunks - 1) // num_chunks
    last = max(size - chunk * (num_chunks - 1), 0)
    if last == 0:
        chunk = size // num_chunks
        last = size - chunk * (num_chunks - 1)
    return [chunk] * (num_chunks - 1) + [last]


def split_tensor_along_dim(tensor, dim, num_chunks):
    assert dim < tensor.dim(), f"Error, tensor dimension is {tensor.dim()} which cannot be split along {dim}"
    assert tensor.shape[dim] >= num_chunks, f"Error, cannot split dim {dim} of size {tensor.shape[dim]} into {num_chunks} chunks"
    return torch.split(tensor, compute_split_shapes(tensor.shape[dim], num_chunks), dim=dim)


# ----------------------------------------------------------------------------
# distributed transpose
# ----------------------------------------------------------------------------
def _transpose(x, dim0, dim1, dim1_split_sizes, group):
    """Split ``dim0`` over the group, gather ``dim1``: one packed all_to_all_single."""
    size = dist.get_world_size(group=group)
    rank = dist.get_rank(group=group)
    dim0 %= x.dim()
    dim1 %= x.dim()
    dim0_split_sizes = compute_split_shapes(x.shape[dim0], size)
    chunks = torch.split(x, dim0_split_sizes, dim=dim0)
    in_splits = [c.numel() for c in chunks]
    if dim0 == 0 and x.is_contiguous():
        send = x.view(-1)        # splitting the outermost axis: the peers' chunks already lie back to back
    else:
        # pack: ONE strided copy per peer straight into the contiguous send buffer
        send = torch.empty(sum(in_splits), dtype=x.dtype, device=x.device)
        off = 0
        for c, n in zip(chunks, in_splits):
            send[off:off + n].view(c.shape).copy_(c)
            off += n
    shp = list(chunks[rank].shape)
    out_shapes = []
    for s in dim1_split_sizes:
        o = list(shp)
        o[dim1] = s
        out_shapes.append(o)
    out_splits = [int(torch.Size(o).numel()) for o in out_shapes]
    recv = torch.empty(sum(out_splits), dtype=x.dtype, device=x.device)
    if x.is_complex():  # RCCL / gloo move bytes; complex64 is viewed as float pairs
        dist.all_to_all_single(torch.view_as_real(recv).view(-1), torch.view_as_real(send).view(-1),
                               [2 * s for s in out_splits], [2 * s for s in in_splits], group=group)
    else:
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=group)
    if dim1 == 0:                # gathering the outermost axis: the received chunks ARE the result
        o = list(shp)
        o[0] = sum(dim1_split_sizes)
        return recv.view(o), dim0_split_sizes
    parts = [p.view(o) for p, o in zip(torch.split(recv, out_splits), out_shapes)]
    return torch.cat(parts, dim=dim1), dim0_split_sizes


class _DistributedTranspose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dims, dim1_split_sizes, group_name):
        group = comm.get_group(group_name)
        out, dim0_split_sizes = _transpose(x, dims[0], dims[1], dim1_split_sizes, group)
        ctx.dims, ctx.dim0_split_sizes, ctx.group = dims, dim0_split_sizes, group   # backward stays on the forward's lane
        return out

    @staticmethod
    def backward(ctx, go):
        gi, _ = _transpose(go.contiguous(), ctx.dims[1], ctx.dims[0], ctx.dim0_split_sizes, ctx.group)
        return gi, None, None, None


class _AllToAllFlat(torch.autograd.Function):
    """all_to_all_single on a flat buffer whose peer chunks already lie back to back (peer-major Fourier rows on one
    side, the outermost axis on the other): no pack, no concatenate; backward = the reverse exchange."""

    @staticmethod
    def forward(ctx, x, in_splits, out_splits, group):
        ctx.cfg = (in_splits, out_splits, group, tuple(x.shape))
        send = x.contiguous().view(-1)
        recv = torch.empty(sum(out_splits), dtype=x.dtype, device=x.device)
        dist.all_to_all_single(torch.view_as_real(recv).view(-1), torch.view_as_real(send).view(-1),
                               [2 * s for s in out_splits], [2 * s for s in in_splits], group=group)
        return recv

    @staticmethod
    def backward(ctx, g):
        in_splits, out_splits, group, shape = ctx.cfg
        send = g.contiguous().view(-1)
        recv = torch.empty(sum(in_splits), dtype=g.dtype, device=g.device)
        dist.all_to_all_single(torch.view_as_real(recv).view(-1), torch.view_as_real(send).view(-1),
                               [2 * s for s in in_splits], [2 * s for s in out_splits], group=group)
        return recv.view(shape), None, None, None


class distributed_transpose_polar:
    """``distributed_transpose_polar.apply(x, (dim0, dim1), dim1_split_sizes)`` over the ``h`` group."""

    @staticmethod
    def apply(x, dims, dim1_split_sizes):
        return _DistributedTranspose.apply(x, tuple(dims), list(dim1_split_sizes), "h")


class distributed_transpose_azimuth:
    """Same over the ``w`` group."""

    @staticmethod
    def apply(x, dims, dim1_split_sizes):
        return _DistributedTranspose.apply(x, tuple(dims), list(dim1_split_sizes), "w")


# ----------------------------------------------------------------------------
# distributed transforms
# ----------------------------------------------------------------------------
class _DistSHTBase(_SHTBase):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self.comm_size_polar = comm.get_size("h")
        self.comm_rank_polar = comm.get_rank("h")
        self.comm_size_azimuth = comm.get_size("w")
        self.comm_rank_azimuth = comm.get_rank("w")
        self.lat_shapes = compute_split_shapes(self.nlat, self.comm_size_polar)
        self.lon_shapes = compute_split_shapes(self.nlon, self.comm_size_azimuth)
        self.l_shapes = compute_split_shapes(self.lmax, self.comm_size_polar)
        self.m_shapes = compute_split_shapes(self.mmax, self.comm_size_azimuth)
        self.l_off = sum(self.l_shapes[: self.comm_rank_polar])
        self.m_off = sum(self.m_shapes[: self.comm_rank_azimuth])
        self.nlat_local = self.lat_shapes[self.comm_rank_polar]
        self.nlon_local = self.lon_shapes[self.comm_rank_azimuth]
        self.lmax_local = self.l_shapes[self.comm_rank_polar]
        self.mmax_local = self.m_shapes[self.comm_rank_azimuth]


class DistributedRealSHT(_DistSHTBase):
    """local real [B, C, nlat_loc, nlon_loc] -> local complex64 [B, C, l_loc, m_loc]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self.register_buffer("weights", ops.legendre_table(grid, nlat, self.lmax, self.mmax, True), persistent=False)

    def forward_packed(self, x):
        """x [B, C, nlat_loc, nlon_loc] -> spectrum [l_loc, m_loc, B*C] (private layout)."""
        B, C = x.shape[0], x.shape[1]
        if self.comm_size_azimuth == 1 and self.comm_size_polar > 1 and ops.SPECTRAL_GEMM == "bf16x3":
            # latitude-major Fourier rows [K, M, B, C]: the latitude all-to-all gathers the OUTERMOST axis, so the
            # received chunks are the Legendre operand as they arrive (no concatenate copy)
            h = self.comm_size_polar
            if C % h == 0 and ops.fft_pm_supported(self.nlon, self.mmax, C, C // h):
                # peer-major rows [h, K_loc, M, B, C/h] ARE the send buffer; the received latitude chunks are the operand
                Ch, kl, unit = C // h, x.shape[2], self.mmax * B * (C // h)
                xf = ops.rfft_pm(x.reshape(B * C, kl, self.nlon).contiguous(), self.twiddles, self.mmax, C, Ch)
                xf = _AllToAllFlat.apply(xf, [kl * unit] * h, [k * unit for k in self.lat_shapes], comm.get_group("h"))
                xf = xf.view(self.nlat, self.mmax, B, Ch)
            else:
                xf = ops.rfft(x.reshape(B * C, x.shape[2], self.nlon).contiguous(), self.twiddles, self.mmax, True)
                xf = xf.view(-1, self.mmax, B, C)                                     # [K_loc, M, B, C]
                xf = distributed_transpose_polar.apply(xf, (3, 0), self.lat_shapes)    # [K, M, B, C_h]
            Ch = xf.shape[3]
            c = ops.legendre_fwd(xf.reshape(self.nlat, self.mmax, B * Ch), self.weights, self.lmax, self.m_off, True)
            c = c.view(self.lmax, -1, B, Ch)
            c = distributed_transpose_polar.apply(c, (0, 3), compute_split_shapes(C, self.comm_size_polar))
            return c.reshape(c.shape[0], c.shape[1], B * C).contiguous()
        if self.comm_size_azimuth > 1:      # make longitude local, split channels over w
            x = distributed_transpose_azimuth.apply(x, (1, -1), self.lon_shapes)
        Cw = x.shape[1]
        xf = ops.rfft(x.reshape(B * Cw, x.shape[2], self.nlon).contiguous(), self.twiddles, self.mmax)
        xf = xf.view(self.mmax, -1, B, Cw)  # [M, K_loc, B, C_w]
        if self.comm_size_azimuth > 1:      # split modes over w, channels local again
            xf = distributed_transpose_azimuth.apply(xf, (0, 3), compute_split_shapes(C, self.comm_size_azimuth))
        if self.comm_size_polar > 1:        # make latitude local, split channels over h
            xf = distributed_transpose_polar.apply(xf, (3, 1), self.lat_shapes)
        Ch = xf.shape[3]
        c = ops.legendre_fwd(xf.reshape(xf.shape[0], self.nlat, B * Ch).contiguous(), self.weights, self.lmax, self.m_off)
        c = c.view(self.lmax, -1, B, Ch)    # [L, M_loc, B, C_h]
        if self.comm_size_polar > 1:        # split degrees over h, channels local again
            c = distributed_transpose_polar.apply(c, (0, 3), compute_split_shapes(C, self.comm_size_polar))
        return c.reshape(c.shape[0], c.shape[1], B * C).contiguous()

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("DistributedRealSHT expects [B, C, nlat_loc, nlon_loc]")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        B, C = x.shape[:2]
        c = ops.spec_unpack(self.forward_packed(x.contiguous()), self.l_off, self.m_off)
        return c.reshape(B, C, self.lmax_local, self.mmax_local)


class DistributedInverseRealSHT(_DistSHTBase):
    """local complex64 [B, C, l_loc, m_loc] -> local real [B, C, nlat_loc, nlon_loc]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self.register_buffer("pct", ops.legendre_table(grid, nlat, self.lmax, self.mmax, False), persistent=False)

    def inverse_packed(self, c, B, out_dtype=torch.float32, want_row_sums=False):
        """spectrum [l_loc, m_loc, B*C] -> x [B, C, nlat_loc, nlon_loc]; ``want_row_sums``: (x, LOCAL fp64 ``[B*C, 2]`` row sums of x
        or None where the path has no kernel for them) -- the sharded instance norm all-reduces them (layer_norm.py)."""
        if want_row_sums:
            x = self._inverse_packed(c, B, out_dtype, True)
            return x if isinstance(x, tuple) else (x, None)
        return self._inverse_packed(c, B, out_dtype, False)

    def _inverse_packed(self, c, B, out_dtype, want_row_sums):
        C = c.shape[2] // B
        c = c.view(c.shape[0], c.shape[1], B, C)
        if self.comm_size_azimuth == 1 and self.comm_size_polar > 1 and ops.SPECTRAL_GEMM == "bf16x3":
            # latitude-major Fourier rows: the latitude all-to-all splits the OUTERMOST axis (no pack copy)
            c = distributed_transpose_polar.apply(c, (3, 0), self.l_shapes)
            Ch = c.shape[3]
            xf = ops.legendre_inv(c.reshape(self.lmax, c.shape[1], B * Ch).contiguous(), self.pct, self.nlat, self.m_off, True)
            xf = xf.view(self.nlat, -1, B, Ch)                                    # [K, M, B, C_h]
            h = self.comm_size_polar
            if C == Ch * h and ops.fft_pm_supported(self.nlon, self.mmax, C, Ch):
                # the latitude chunks are sent as they lie; the received buffer [h, K_loc, M, B, C/h] is read peer-major
                kl, unit = self.nlat_local, self.mmax * B * Ch
                xf = _AllToAllFlat.apply(xf, [k * unit for k in self.lat_shapes], [kl * unit] * h, comm.get_group("h"))
                if want_row_sums and ops.irfft_sums_supported(self.nlon, self.mmax) and out_dtype in (torch.float32, torch.bfloat16):
                    x, sums = ops.irfft_pm(xf.view(h, kl, self.mmax, B * Ch), self.twiddles, self.nlon, out_dtype, C, Ch, True)
                    return x.view(B, C, -1, self.nlon), sums
                x = ops.irfft_pm(xf.view(h, kl, self.mmax, B * Ch), self.twiddles, self.nlon, out_dtype, C, Ch)
                return x.view(B, C, -1, self.nlon)
            xf = distributed_transpose_polar.apply(xf, (0, 3), compute_split_shapes(C, self.comm_size_polar))
            x = ops.irfft(xf.reshape(xf.shape[0], self.mmax, B * C).contiguous(), self.twiddles, self.nlon, out_dtype, True)
            return x.view(B, C, -1, self.nlon)
        if self.comm_size_polar > 1:        # make degrees local, split channels over h
            c = distributed_transpose_polar.apply(c, (3, 0), self.l_shapes)
        Ch = c.shape[3]
        xf = ops.legendre_inv(c.reshape(self.lmax, c.shape[1], B * Ch).contiguous(), self.pct, self.nlat, self.m_off)
        xf = xf.view(-1, self.nlat, B, Ch)  # [M_loc, K, B, C_h]
        if self.comm_size_polar > 1:        # split latitude over h, channels local again
            xf = distributed_transpose_polar.apply(xf, (1, 3), compute_split_shapes(C, self.comm_size_polar))
        if self.comm_size_azimuth > 1:      # make modes local, split channels over w
            xf = distributed_transpose_azimuth.apply(xf, (3, 0), self.m_shapes)
        Cw = xf.shape[3]
        x = ops.irfft(xf.reshape(self.mmax, xf.shape[1], B * Cw).contiguous(), self.twiddles, self.nlon, out_dtype)
        x = x.view(B, Cw, -1, self.nlon)
        if self.comm_size_azimuth > 1:      # split longitude over w, channels local again
            x = distributed_transpose_azimuth.apply(x, (-1, 1), compute_split_shapes(C, self.comm_size_azimuth))
        return x

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("DistributedInverseRealSHT expects [B, C, l_loc, m_loc]")
        if x.dtype != torch.complex64:
            x = x.to(torch.complex64)
        B, C = x.shape[:2]
        c = ops.spec_pack(x.reshape(B * C, self.lmax_local, self.mmax_local).contiguous(), self.l_off, self.m_off)
        return self.inverse_packed(c, B)


# ----------------------------------------------------------------------------
# distributed planar transforms (the FNO variant of the network: spectral_transform="fft" under spatial parallelism)
# ----------------------------------------------------------------------------
class _DistFFT2Base(nn.Module):
    """Sizes and shard shapes of ``makani/mpu/layers.py:38-66,105-133`` (same attribute names)."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None):
        super().__init__()
        self.comm_size_h, self.comm_size_w = comm.get_size("h"), comm.get_size("w")
        self.comm_rank_w = comm.get_rank("w")
        self.nlat, self.nlon = nlat, nlon
        self.lmax = min(lmax or self.nlat, self.nlat)
        self.mmax = min(mmax or self.nlon // 2 + 1, self.nlon // 2 + 1)
        self.lmax_high, self.lmax_low = math.ceil(self.lmax / 2), math.floor(self.lmax / 2)
        self.lat_shapes = compute_split_shapes(self.nlat, self.comm_size_h)
        self.lon_shapes = compute_split_shapes(self.nlon, self.comm_size_w)
        self.l_shapes = compute_split_shapes(self.lmax, self.comm_size_h)
        self.m_shapes = compute_split_shapes(self.mmax, self.comm_size_w)


class DistributedRealFFT2(_DistFFT2Base):
    """``mpu/layers.py:38-100``: x [B, C, H/h, W/w] real -> [B, C, lmax/h, mmax/w] complex, ``rfft2(norm="ortho")`` with the
    two axes made local one after the other by channel <-> axis transposes (one packed all-to-all each; the local FFTs are
    torch.fft -- the planar transform is not on the SFNO hot path)."""

    def forward(self, x):
        num_chans = x.shape[1]
        if self.comm_size_w > 1:                                              # w local, channels split
            x = distributed_transpose_azimuth.apply(x, (1, -1), self.lon_shapes)
        x = torch.fft.rfft(x, n=self.nlon, dim=-1, norm="ortho")[..., :self.mmax].contiguous()
        if self.comm_size_w > 1:                                              # m split, channels local
            x = distributed_transpose_azimuth.apply(x, (-1, 1), compute_split_shapes(num_chans, self.comm_size_w))
        if self.comm_size_h > 1:                                              # h local, channels split
            x = distributed_transpose_polar.apply(x, (1, -2), self.lat_shapes)
        x = torch.fft.fft(x, n=self.nlat, dim=-2, norm="ortho")
        x = torch.cat([x[..., :self.lmax_high, :], x[..., -self.lmax_low:, :]], dim=-2)
        if self.comm_size_h > 1:                                              # l split, channels local
            x = distributed_transpose_polar.apply(x, (-2, 1), compute_split_shapes(num_chans, self.comm_size_h))
        return x


class DistributedInverseRealFFT2(_DistFFT2Base):
    """``mpu/layers.py:103-169``: the inverse of the above (zero padding between the high and low latitude modes when
    ``lmax < nlat``)."""

    def forward(self, x):
        num_chans = x.shape[1]
        if self.comm_size_h > 1:                                              # l local, channels split
            x = distributed_transpose_polar.apply(x, (1, -2), self.l_shapes)
        if self.lmax < self.nlat:
            xh, xl = x[..., :self.lmax_high, :], x[..., -self.lmax_low:, :]
            x = torch.cat([torch.nn.functional.pad(xh, (0, 0, 0, self.nlat - self.lmax)), xl], dim=-2)
        x = torch.fft.ifft(x, n=self.nlat, dim=-2, norm="ortho")
        if self.comm_size_h > 1:                                              # h split, channels local
            x = distributed_transpose_polar.apply(x, (-2, 1), compute_split_shapes(num_chans, self.comm_size_h))
        if self.comm_size_w > 1:                                              # m local, channels split
            x = distributed_transpose_azimuth.apply(x, (1, -1), self.m_shapes)
        x = torch.fft.irfft(x, n=self.nlon, dim=-1, norm="ortho")
        if self.comm_size_w > 1:                                              # w split, channels local
            x = distributed_transpose_azimuth.apply(x, (-1, 1), compute_split_shapes(num_chans, self.comm_size_w))
        return x
