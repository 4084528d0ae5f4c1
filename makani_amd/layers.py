"""Glue layers that run between the spectral ops: encoder/decoder, MLP, DropPath, FFT wrappers.

Same constructor arguments, initialisation and ``state_dict`` keys as
``makani/models/common/layers.py:35-287``.  Pointwise 1x1 convolutions are evaluated as
one GEMM over the NCHW field viewed as ``[C, H*W]`` (hipBLASLt through torch) -- the
parameters stay ``nn.Conv2d`` weights ``[O, I, 1, 1]`` so reference checkpoints load.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint


class _PointwiseConv(torch.autograd.Function):
    """y[b] = W @ x[b] on the NCHW field viewed as [C, H*W]; hand-written backward.

    hipBLASLt runs the weight-gradient GEMM  gW = gY @ X^T  (contraction over H*W ~ 1e5..1e6,
    output only [O x I]) on a few dozen workgroups; splitting the contraction into SPLIT batched
    chunks (strided views, no copies) and summing the partials in fp32 fills the chip.
    """
    SPLIT = int(__import__("os").environ.get("MK_WGRAD_SPLIT", "32"))

    @staticmethod
    def forward(ctx, x3, w_master, addend):
        # x3 [B, I, P], w_master [O, I] (the parameter view, any float dtype), optional addend [B, O, P] (skip
        # connection folded into the GEMM epilogue: y = addend + W x).  The weight is cast to the activation dtype
        # here, inside the node, so its gradient goes back in fp32 straight from the wgrad kernel (no bf16 round
        # trip, two tiny cast kernels fewer per convolution).
        w = w_master if w_master.dtype == x3.dtype else w_master.to(x3.dtype)
        ctx.master_dtype = w_master.dtype
        ctx.save_for_backward(x3, w)
        ctx.has_addend = addend is not None
        if addend is not None:
            # y = addend + W x accumulated IN PLACE into the addend's storage: an out-of-place addmm first copies the
            # addend into the result (one D2D copy of the whole activation per block, 0.5 ms per step).  The addend is
            # the block's norm1 output, which no backward needs (the norm keeps its input and statistics, this node
            # returns gy for it); should anyone have saved it, autograd's version check fails loudly in backward.
            out = addend.detach()
            if x3.shape[0] == 1:
                torch.addmm(out[0], w, x3[0], out=out[0])
            else:
                torch.baddbmm(out, w.unsqueeze(0).expand(x3.shape[0], -1, -1), x3, out=out)
            return out
        if x3.shape[0] == 1:
            return torch.mm(w, x3[0]).unsqueeze(0)
        return torch.bmm(w.unsqueeze(0).expand(x3.shape[0], -1, -1), x3)

    @staticmethod
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        B, I, P = x3.shape
        O = w.shape[0]
        gy = gy.contiguous()
        gx = gw = None
        ga = gy if ctx.has_addend else None
        if ctx.needs_input_grad[0]:
            if B == 1:
                gx = torch.mm(w.t(), gy[0]).unsqueeze(0)
            else:
                gx = torch.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gy)
        if ctx.needs_input_grad[1] and x3.is_cuda and x3.dtype == torch.bfloat16 and gy.dtype == torch.bfloat16 \
                and P % 8 == 0 and os.environ.get("MK_CONV_WGRAD", "hip") == "hip":
            # hand-written bf16 MFMA kernel: pixel slabs per workgroup, fp32 atomics into gW
            from . import ops
            gw = ops.conv1x1_wgrad_raw(gy, x3.contiguous()).to(ctx.master_dtype)
        elif ctx.needs_input_grad[1]:
            S = _PointwiseConv.SPLIT
            while S > 1 and P % S:
                S //= 2
            kc = P // S
            # per sample: [O, S, kc] -> [S, O, kc] and [I, S, kc] -> [S, kc, I] are strided views (no copies)
            gw32 = None
            for bi in range(B):
                a = gy[bi].view(O, S, kc).permute(1, 0, 2)
                b = x3[bi].view(I, S, kc).permute(1, 2, 0)
                part = torch.bmm(a, b).float().sum(0)
                gw32 = part if gw32 is None else gw32 + part
            gw = gw32.to(ctx.master_dtype)
        return gx, gw, ga


class Conv1x1(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 1)`` evaluated as ``W @ x.view(B, C, H*W)`` (+ bias)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(in_channels, out_channels, 1, bias=bias)

    def forward(self, x, add_bias=True, addend=None):
        """``addend`` (same shape as the output) is added inside the GEMM epilogue: y = addend + conv(x)."""
        if x.dim() != 4 or not x.is_contiguous() or not x.is_cuda:
            y = F.conv2d(x, self.weight, self.bias if add_bias else None)
            return y if addend is None else y + addend
        B, C, H, W = x.shape
        w = self.weight.view(self.out_channels, self.in_channels)
        # NOTE: torch.matmul(2-D, 3-D) folds through a transposed *copy* of the activation; mm / bmm on
        # the [C, H*W] row-major view go straight to hipBLASLt with no copy in forward or backward
        if torch.is_autocast_enabled():
            x3 = x.view(B, C, H * W).to(torch.get_autocast_dtype('cuda'))
        else:
            x3 = x.view(B, C, H * W)
        with torch.autocast("cuda", enabled=False):
            a3 = None
            if addend is not None:
                a3 = addend.contiguous().view(B, self.out_channels, H * W).to(x3.dtype)
            y = _PointwiseConv.apply(x3, w, a3)
        if self.bias is not None and add_bias:
            y = y + self.bias.to(y.dtype).view(1, -1, 1)
        return y.view(B, self.out_channels, H, W)


def _is_exact_gelu(m):
    return isinstance(m, nn.GELU) and getattr(m, "approximate", "none") == "none"


def run_pointwise_chain(mods, x, skip_last_bias=False):
    """Evaluate an ``nn.Sequential`` of 1x1 convs / activations / identities, fusing every
    ``Conv1x1(bias) -> GELU`` pair into one HIP bias+GELU pass over the conv output.

    ``skip_last_bias``: the caller feeds the result straight into an instance norm, which removes any
    per-channel constant -- the last conv's bias add is then a no-op on the output (and its gradient is
    identically zero), so the pass over the tensor is skipped.
    """
    from . import ops
    mods = list(mods)
    last_conv = max((j for j, m in enumerate(mods) if isinstance(m, Conv1x1)), default=-1)
    tail_is_identity = all(isinstance(m, nn.Identity) for m in mods[last_conv + 1:])
    i = 0
    while i < len(mods):
        m = mods[i]
        if skip_last_bias and tail_is_identity and i == last_conv and x.is_cuda:
            x = m(x, add_bias=False)
            i += 1
            continue
        if isinstance(m, Conv1x1) and i + 1 < len(mods) and _is_exact_gelu(mods[i + 1]) and x.is_cuda:
            y = m(x, add_bias=False)
            if ops.pointwise_supported(y):
                x = ops.bias_gelu(y, m.bias)
            else:
                if m.bias is not None:
                    y = y + m.bias.to(y.dtype).view(1, -1, 1, 1)
                x = mods[i + 1](y)
            i += 2
        else:
            x = m(x)
            i += 1
    return x


class InstanceNorm2d(nn.InstanceNorm2d):
    """``nn.InstanceNorm2d(affine=True, track_running_stats=False)`` as two HIP streaming passes
    (row sums, apply), optionally with the block's GELU fused into the apply pass."""

    def forward(self, x, fuse_gelu=False):
        from . import ops
        if self.track_running_stats or not ops.pointwise_supported(x):
            y = super().forward(x)
            return F.gelu(y) if fuse_gelu else y
        return ops.instance_norm(x, self.weight, self.bias, self.eps, fuse_gelu)


def drop_path(x, drop_prob=0.0, training=False):
    if drop_prob == 0.0 or not training:
        return x
    keep_prob = 1.0 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
    random_tensor.floor_()
    return x.div(keep_prob) * random_tensor


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class EncoderDecoder(nn.Module):
    """layers.py:86-133."""

    def __init__(self, num_layers, input_dim, output_dim, hidden_dim, act_layer, gain=1.0, input_format="nchw"):
        super().__init__()
        if input_format not in ("nchw", "traditional"):
            raise NotImplementedError(f"Error, input format {input_format} not supported.")
        mods = []
        current_dim = input_dim
        for _ in range(num_layers):
            if input_format == "nchw":
                mods.append(Conv1x1(current_dim, hidden_dim, bias=True))
            else:
                mods.append(nn.Linear(current_dim, hidden_dim, bias=True))
            mods[-1].weight.is_shared_mp = ["spatial"]
            nn.init.normal_(mods[-1].weight, mean=0.0, std=math.sqrt(2.0 / current_dim))
            if mods[-1].bias is not None:
                mods[-1].bias.is_shared_mp = ["spatial"]
                nn.init.constant_(mods[-1].bias, 0.0)
            mods.append(act_layer())
            current_dim = hidden_dim
        if input_format == "nchw":
            mods.append(Conv1x1(current_dim, output_dim, bias=False))
        else:
            mods.append(nn.Linear(current_dim, output_dim, bias=False))
        mods[-1].weight.is_shared_mp = ["spatial"]
        nn.init.normal_(mods[-1].weight, mean=0.0, std=math.sqrt(gain / current_dim))
        self.fwd = nn.Sequential(*mods)

    def forward(self, x):
        return run_pointwise_chain(self.fwd, x)


class MLP(nn.Module):
    """layers.py:136-216."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, output_bias=True,
                 input_format="nchw", drop_rate=0.0, drop_type="iid", checkpointing=0, gain=1.0, **kwargs):
        super().__init__()
        self.checkpointing = checkpointing
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if input_format == "nchw":
            fc1 = Conv1x1(in_features, hidden_features, bias=True)
            fc1.weight.is_shared_mp = ["spatial"]
            fc1.bias.is_shared_mp = ["spatial"]
        elif input_format == "traditional":
            fc1 = nn.Linear(in_features, hidden_features, bias=True)
        else:
            raise NotImplementedError(f"Error, input format {input_format} not supported.")
        nn.init.normal_(fc1.weight, mean=0.0, std=math.sqrt(2.0 / in_features))
        nn.init.constant_(fc1.bias, 0.0)
        act = act_layer()
        if (input_format == "traditional") and (drop_type == "features"):
            raise NotImplementedError("Error, traditional input format and feature dropout cannot be selected simultaneously")
        if input_format == "nchw":
            fc2 = Conv1x1(hidden_features, out_features, bias=output_bias)
            fc2.weight.is_shared_mp = ["spatial"]
            if output_bias:
                fc2.bias.is_shared_mp = ["spatial"]
        else:
            fc2 = nn.Linear(hidden_features, out_features, bias=output_bias)
        nn.init.normal_(fc2.weight, mean=0.0, std=math.sqrt(gain / hidden_features))
        if fc2.bias is not None:
            nn.init.constant_(fc2.bias, 0.0)
        if drop_rate > 0.0:
            if drop_type == "iid":
                drop = nn.Dropout(drop_rate)
            elif drop_type == "features":
                drop = nn.Dropout2d(drop_rate)
            else:
                raise NotImplementedError(f"Error, drop_type {drop_type} not supported")
        else:
            drop = nn.Identity()
        self.fwd = nn.Sequential(fc1, act, drop, fc2, drop)

    def _run(self, x, skip_last_bias=False):
        return run_pointwise_chain(self.fwd, x, skip_last_bias)

    def checkpoint_forward(self, x, skip_last_bias=False):
        return checkpoint(self._run, x, skip_last_bias, use_reentrant=False)

    def forward(self, x, skip_last_bias=False):
        if self.checkpointing >= 2:
            return self.checkpoint_forward(x, skip_last_bias)
        return self._run(x, skip_last_bias)


class RealFFT2(nn.Module):
    """layers.py:219-250 -- the duck-typed planar transform (torch.fft; not on the SFNO path)."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None):
        super().__init__()
        self.nlat, self.nlon = nlat, nlon
        self.lmax = min(lmax or self.nlat, self.nlat)
        self.mmax = min(mmax or self.nlon // 2 + 1, self.nlon // 2 + 1)
        self.truncate = not ((self.lmax == self.nlat) and (self.mmax == (self.nlon // 2 + 1)))
        self.lmax_high = math.ceil(self.lmax / 2)
        self.lmax_low = math.floor(self.lmax / 2)

    def forward(self, x):
        y = torch.fft.rfft2(x, s=(self.nlat, self.nlon), dim=(-2, -1), norm="ortho")
        if self.truncate:
            y = torch.cat((y[..., : self.lmax_high, : self.mmax], y[..., -self.lmax_low:, : self.mmax]), dim=-2)
        return y


class InverseRealFFT2(nn.Module):
    """layers.py:253-287."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None):
        super().__init__()
        self.nlat, self.nlon = nlat, nlon
        self.lmax = min(lmax or self.nlat, self.nlat)
        self.mmax = min(mmax or self.nlon // 2 + 1, self.nlon // 2 + 1)
        self.truncate = not ((self.lmax == self.nlat) and (self.mmax == (self.nlon // 2 + 1)))
        self.lmax_high = math.ceil(self.lmax / 2)
        self.lmax_low = math.floor(self.lmax / 2)

    def forward(self, x):
        xt = x[..., : self.mmax]
        if self.truncate:
            xth = xt[..., : self.lmax_high, :]
            xtl = xt[..., -self.lmax_low:, :]
            xthp = F.pad(xth, (0, 0, 0, self.nlat - self.lmax))
            xt = torch.cat([xthp, xtl], dim=-2)
        return torch.fft.irfft2(xt, s=(self.nlat, self.nlon), dim=(-2, -1), norm="ortho")
