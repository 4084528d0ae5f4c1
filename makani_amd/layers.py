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


class _X3Conv(torch.autograd.Function):
    """y[b] = W @ x[b] (+ addend[b], accumulated in place) on fp32 ``[B, C, P]`` fields: forward, data gradient and weight
    gradient on the bf16x3 MFMA engine of the spectral GEMMs (``mk_conv1x1_x3``: fp32-accurate, hand-written) -- the path of
    every 1x1 convolution outside bf16 autocast."""

    @staticmethod
    def forward(ctx, x3, w, addend):
        from . import ops
        wf = w.detach().float()
        ctx.save_for_backward(x3, wf)
        ctx.cfg = (w.dtype, addend is not None)
        return ops.conv1x1_x3(wf, x3, out=None if addend is None else addend.detach())

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x3, wf = ctx.saved_tensors
        wdt, has_addend = ctx.cfg
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = ops.conv1x1_x3(wf.t().contiguous(), gy)
        if ctx.needs_input_grad[1]:
            gw = ops.conv1x1_x3_wgrad(gy, x3).to(wdt)
        return gx, gw, (gy if has_addend else None)


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


def _step_items(w):
    """What the backward pass of a convolution with weight ``w`` takes from the open ``ops.EngineArena`` scope: (transposed
    image, zeroed gradient buffer), each None outside a scope (the backward pass then packs / allocates itself)."""
    from . import ops
    return ops.arena_image(w, True), (ops.arena_grad_buffer(w) if w.requires_grad else None)


def _row_sums(t3):
    """sum over (batch, pixels) of a [B, C, P] field -> fp32 [C] (the HIP row-sum pass of the instance norm)."""
    from . import ops
    r = ops.row_sums(t3).view(t3.shape[0], t3.shape[1])
    return r[0] if r.shape[0] == 1 else r.sum(0)      # one batch item: nothing to add up (saves a launch per bias gradient)


class _PceConv(torch.autograd.Function):
    """y[b] = W @ x[b] (+ bias) (+ addend[b]) on bf16 ``[B, C, P]`` fields, one pixel-column-engine launch
    (csrc/pce.hip) each way; the weight gradient comes from the bf16 MFMA wgrad kernel in fp32."""

    @staticmethod
    def forward(ctx, x3, w, bias, addend):
        from . import ops
        ctx.save_for_backward(x3, w)
        ctx.has = (bias is not None, addend is not None)
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.step = _step_items(w)
        return ops.pce_gemm(x3, ops.pce_pack(w), w.shape[0], bias=bias, addend=addend)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x3, w = ctx.saved_tensors
        has_bias, has_addend = ctx.has
        gy = gy.contiguous()
        gx = gw = gb = None
        # taken once: no reference may survive this call (autograd adopts gw only as its sole owner); a second backward pass
        # through the same node packs / allocates for itself
        (wimg_t, gbuf), ctx.step = (ctx.step or (None, None)), None
        if ctx.needs_input_grad[0]:
            gx = ops.pce_gemm(gy, wimg_t if wimg_t is not None else ops.pce_pack(w, transpose=True), w.shape[1])
        if ctx.needs_input_grad[1]:
            gw = ops.conv1x1_wgrad_raw(gy, x3, out=gbuf).to(w.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _row_sums(gy).to(ctx.bias_dtype)
        return gx, gw, gb, (gy if has_addend else None)


class _PceConvNormAdd(torch.autograd.Function):
    """out[b] = W @ x[b] (+ bias) + instance_norm(z)[b]: the tail of an SFNO block (``sfnonet.py:262-267``: norm1 of the
    MLP output, then the outer skip) as ONE engine launch.  The norm's statistics come with ``z`` from the launch that
    produced it (``sums``: fp64 ``[B * C, 2]`` local row sums), a 2 us kernel turns them into per-row coefficients, and the
    GEMM epilogue applies ``a * z + b`` while it adds the skip -- the normalised field is never written.  Backward: the
    convolution's two gradients as in ``_PceConv``; the norm's from ``z``, the incoming gradient and the saved (mean, rstd)."""

    @staticmethod
    def forward(ctx, x3, w, bias, z4, sums, nw, nb, eps, group, count):
        from . import ops
        B, C, H, W = z4.shape
        if group is not None:
            torch.distributed.all_reduce(sums, group=group)
        cnt = H * W if group is None else int(count)
        wf = None if nw is None else nw.detach().float().contiguous()
        bf = None if nb is None else nb.detach().float().contiguous()
        stats, affine = ops.instance_norm_coeffs(sums, wf, bf, B * C, C, cnt, eps)
        y = ops.pce_gemm(x3, ops.pce_pack(w), w.shape[0], bias=bias, addend=z4.view(B, C, H * W), addend_affine=affine)
        empty = z4.new_empty(0, dtype=torch.float32)
        ctx.save_for_backward(x3, w, z4, stats, wf if wf is not None else empty, bf if bf is not None else empty)
        ctx.cfg = (None if bias is None else bias.dtype, None if nw is None else nw.dtype, None if nb is None else nb.dtype,
                   group, cnt)
        ctx.step = _step_items(w)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x3, w, z4, stats, wf, bf = ctx.saved_tensors
        bias_dtype, nw_dtype, nb_dtype, group, cnt = ctx.cfg
        B, C, H, W = z4.shape
        gy = gy.contiguous()
        gx = gw = gb = None
        # taken once: no reference may survive this call (autograd adopts gw only as its sole owner); a second backward pass
        # through the same node packs / allocates for itself
        (wimg_t, gbuf), ctx.step = (ctx.step or (None, None)), None
        if ctx.needs_input_grad[0]:
            gx = ops.pce_gemm(gy, wimg_t if wimg_t is not None else ops.pce_pack(w, transpose=True), w.shape[1])
        if ctx.needs_input_grad[1]:
            gw = ops.conv1x1_wgrad_raw(gy, x3, out=gbuf).to(w.dtype)
        if bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = _row_sums(gy).to(bias_dtype)
        gz, gnw, gnb = ops.instance_norm_backward(z4, gy.view(B, C, H, W), stats, wf if nw_dtype is not None else None,
                                                  bf if nb_dtype is not None else None, False, group, cnt,
                                                  grad_dtype=nw_dtype if (nw_dtype is not None and nw_dtype == nb_dtype) else None)
        return (gx, gw, gb, gz, None, gnw.to(nw_dtype) if nw_dtype is not None else None,
                gnb.to(nb_dtype) if nb_dtype is not None else None, None, None, None)


def conv_plus_instance_norm(conv, x, z, sums, norm):
    """``conv(x) + norm(z)`` as one launch (``_PceConvNormAdd``) when the engine can take it, else None: ``conv`` a
    ``Conv1x1``, ``norm`` an affine-or-not instance norm on the field's own statistics (single-GPU or spatially
    distributed), ``z`` a bf16 field that came with its row sums ``sums``."""
    from . import comm, ops
    from .layer_norm import DistributedInstanceNorm2d
    if sums is None or not isinstance(conv, Conv1x1) or z.dtype != torch.bfloat16 or not z.is_contiguous() or z.dim() != 4:
        return None
    if os.environ.get("MK_NORM_SKIP_FUSION", "1") == "0":
        return None
    group = count = None
    if isinstance(norm, DistributedInstanceNorm2d):
        if comm.get_size("spatial") > 1:
            group, count = comm.get_group("spatial"), norm._global_count(z)
        nw, nb = (norm.weight, norm.bias) if norm.affine else (None, None)
    elif isinstance(norm, InstanceNorm2d) and not norm.track_running_stats:
        nw, nb = norm.weight, norm.bias
    else:
        return None
    x3 = _engine_field(x)
    if x3 is None or not ops.pce_supported_train(conv.out_channels, conv.in_channels) or conv.out_channels != z.shape[1]:
        return None
    B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    if tuple(z.shape) != (B, conv.out_channels, H, W):
        return None
    with torch.autocast("cuda", enabled=False):
        y = _PceConvNormAdd.apply(x3, conv.weight2d(), conv.bias, z, sums, nw, nb, norm.eps, group, count)
    return y.view(B, conv.out_channels, H, W)


class _PceMLP(torch.autograd.Function):
    """y = W2 @ gelu(W1 @ x + b1) (+ b2): the MLP / encoder / decoder of ``layers.py:86-216`` on bf16 ``[B, C, P]``
    fields.  Forward: two engine launches, bias + GELU in the epilogue of the first (which also keeps the
    pre-activation).  Backward: ``gpre = (W2^T gy) * gelu'(pre)`` in one launch, ``gx = W1^T gpre`` in another, the two
    weight gradients from the wgrad kernel, the bias gradients from row sums.

    ``b2`` may be switched off with ``apply_b2=False`` (the caller normalises the result per channel, which removes any
    per-channel constant): the parameter stays in the graph and receives the exact gradient of that case -- zero -- so
    wrappers that expect every parameter to get a gradient (DistributedDataParallel, ``mpu/mappings.py:86-96``) work."""

    @staticmethod
    def forward(ctx, x3, w1, b1, w2, b2, apply_b2, want_row_sums, keep_pre=True):
        # keep_pre: the caller's `torch.is_grad_enabled()` (inside forward grad mode is always off, and needs_input_grad ignores it)
        from . import ops
        if keep_pre:
            h, pre = ops.pce_gemm(x3, ops.pce_pack(w1), w1.shape[0], bias=b1, want_pre=True, gelu=True)
        else:       # inference: nobody will ask for GELU'(pre) -- the first launch writes the hidden field once, not twice
            h, pre = ops.pce_gemm(x3, ops.pce_pack(w1), w1.shape[0], bias=b1, gelu=True), x3.new_empty(0)
        want_row_sums = bool(want_row_sums) and w2.shape[0] <= 768
        y = ops.pce_gemm(h, ops.pce_pack(w2), w2.shape[0], bias=b2 if (apply_b2 and b2 is not None) else None,
                         want_row_sums=want_row_sums)
        y, sums = y if want_row_sums else (y, x3.new_empty(0, dtype=torch.float64))
        ctx.save_for_backward(x3, w1, w2, pre, h)
        ctx.cfg = (None if b1 is None else b1.dtype, None if b2 is None else (b2.dtype, tuple(b2.shape)), bool(apply_b2))
        ctx.step = _step_items(w1) + _step_items(w2)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, gy, _gsums):
        from . import ops
        x3, w1, w2, pre, h = ctx.saved_tensors
        b1_dtype, b2_info, apply_b2 = ctx.cfg
        gy = gy.contiguous()
        need_gb1 = b1_dtype is not None and ctx.needs_input_grad[2]
        fused_gb1 = need_gb1 and w2.shape[1] <= 768
        (w1img_t, g1buf, w2img_t, g2buf), ctx.step = (ctx.step or (None, None, None, None)), None
        gpre = ops.pce_gemm(gy, w2img_t if w2img_t is not None else ops.pce_pack(w2, transpose=True), w2.shape[1], aux_in=pre,
                            want_row_sums=fused_gb1)
        if fused_gb1:       # the bias gradient is the pixel sum of gpre: a by-product of the launch that wrote it
            gpre, gsum = gpre
        gx = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[0]:
            gx = ops.pce_gemm(gpre, w1img_t if w1img_t is not None else ops.pce_pack(w1, transpose=True), w1.shape[1])
        if ctx.needs_input_grad[1]:
            gw1 = ops.conv1x1_wgrad_raw(gpre, x3, out=g1buf).to(w1.dtype)
        if need_gb1:
            if fused_gb1:
                g1 = gsum.view(gy.shape[0], -1, 2)[..., 0]
                gb1 = (g1[0] if g1.shape[0] == 1 else g1.sum(0)).to(b1_dtype)
            else:
                gb1 = _row_sums(gpre).to(b1_dtype)
        if ctx.needs_input_grad[3]:
            gw2 = ops.conv1x1_wgrad_raw(gy, h, out=g2buf).to(w2.dtype)
        if b2_info is not None and ctx.needs_input_grad[4]:
            gb2 = _row_sums(gy).to(b2_info[0]) if apply_b2 else torch.zeros(b2_info[1], dtype=b2_info[0], device=gy.device)
        return gx, gw1, gb1, gw2, gb2, None, None, None


class _PceMLPFused(torch.autograd.Function):
    """The same node as ``_PceMLP`` on the fused kernel (``mk_pce_mlp``, csrc/pce_mlp.hip): conv -> GELU -> conv in ONE launch with
    the hidden field on chip, forward and backward; only the pre-activation is kept (bf16), the second weight gradient applies
    the GELU while it stages it (``mk_conv1x1_wgrad_act``).  Opt-in (``MK_MLP_FUSED=1``): on MI355X the launch is 8-13 % faster
    than the pair it replaces going forward, 5 % slower going backward, and the activated weight gradient costs more than the
    bytes saved -- the step as a whole does not gain."""

    @staticmethod
    def forward(ctx, x3, w1, b1, w2, b2, apply_b2, want_row_sums):
        from . import ops
        out = ops.pce_mlp(x3, ops.pce_mlp_pack(w1, False, w2, False), 0, b1=b1,
                          b2=b2 if (apply_b2 and b2 is not None) else None, want_row_sums=bool(want_row_sums))
        y, pre = out[0], out[1]
        sums = out[2] if want_row_sums else x3.new_empty(0, dtype=torch.float64)
        ctx.save_for_backward(x3, w1, w2, pre)
        ctx.cfg = (None if b1 is None else b1.dtype, None if b2 is None else (b2.dtype, tuple(b2.shape)), bool(apply_b2))
        ctx.step = (ops.arena_grad_buffer(w1) if w1.requires_grad else None, ops.arena_grad_buffer(w2) if w2.requires_grad else None)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, gy, _gsums):
        from . import ops
        x3, w1, w2, pre = ctx.saved_tensors
        b1_dtype, b2_info, apply_b2 = ctx.cfg
        (g1buf, g2buf), ctx.step = (ctx.step or (None, None)), None
        gy = gy.contiguous()
        need_gb1 = b1_dtype is not None and ctx.needs_input_grad[2]
        out = ops.pce_mlp(gy, ops.pce_mlp_pack(w2, True, w1, True), 1, pre=pre, want_mid_sums=need_gb1)
        gx, gpre = out[0], out[1]
        gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[1]:
            gw1 = ops.conv1x1_wgrad_raw(gpre, x3, out=g1buf).to(w1.dtype)
        if need_gb1:
            g1 = out[2].view(gy.shape[0], -1)
            gb1 = (g1[0] if g1.shape[0] == 1 else g1.sum(0)).to(b1_dtype)
        if ctx.needs_input_grad[3]:
            gw2 = ops.conv1x1_wgrad_raw(gy, pre, x_gelu=True, out=g2buf).to(w2.dtype)
        if b2_info is not None and ctx.needs_input_grad[4]:
            gb2 = _row_sums(gy).to(b2_info[0]) if apply_b2 else torch.zeros(b2_info[1], dtype=b2_info[0], device=gy.device)
        return (gx if ctx.needs_input_grad[0] else None), gw1, gb1, gw2, gb2, None, None


def _mlp_fused_ok(fc1, fc2):
    """The fused node takes the layer in both directions (K1, M <= 384, hidden <= 768) and is switched on."""
    from . import ops
    return (os.environ.get("MK_MLP_FUSED", "0") == "1" and fc2.out_channels <= 384
            and ops.pce_mlp_supported(fc2.out_channels, fc1.out_channels, fc1.in_channels)
            and ops.pce_mlp_supported(fc1.in_channels, fc1.out_channels, fc2.out_channels))


def _engine_field(x):
    """The field as the pixel-column engine takes it (bf16 ``[B, C, P]``, contiguous, P a multiple of 8), or None."""
    if not (x.is_cuda and x.dim() == 4 and x.is_contiguous() and (x.shape[2] * x.shape[3]) % 8 == 0):
        return None
    if os.environ.get("MK_CONV_ENGINE", "pce") != "pce":
        return None
    if x.dtype == torch.bfloat16:
        return x.view(x.shape[0], x.shape[1], -1)
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.dtype == torch.float32:
        return x.view(x.shape[0], x.shape[1], -1).to(torch.bfloat16)
    return None


class Conv1x1(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 1)`` (same parameters and ``state_dict`` entries) evaluated as a GEMM over the pixels."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(in_channels, out_channels, 1, bias=bias)

    def weight2d(self):
        return self.weight.view(self.out_channels, self.in_channels)

    def forward(self, x, add_bias=True, addend=None):
        """``addend`` (same shape as the output) is added inside the GEMM epilogue: y = addend + conv(x)."""
        from . import ops
        B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
        x3 = _engine_field(x) if x.dim() == 4 else None
        if x3 is not None and ops.pce_supported_train(self.out_channels, self.in_channels):
            a3 = None
            if addend is not None:
                a3 = addend.contiguous().view(B, self.out_channels, H * W).to(torch.bfloat16)
            with torch.autocast("cuda", enabled=False):
                y = _PceConv.apply(x3, self.weight2d(), self.bias if add_bias else None, a3)
            return y.view(B, self.out_channels, H, W)
        if x.dim() != 4 or not x.is_contiguous() or not x.is_cuda:
            y = F.conv2d(x, self.weight, self.bias if add_bias else None)
            return y if addend is None else y + addend
        if addend is None and add_bias and self.bias is not None:
            y = _x3_inference_conv(self, x, False)        # fp32 inference: the bias add rides in the GEMM's epilogue
            if y is not None:
                return y
        # fp32 fields (no autocast): the bf16x3 engine (`_X3Conv`); torch mm / bmm only as the MK_CONV_FP32=torch A/B path
        if torch.is_autocast_enabled():
            x3 = x.view(B, x.shape[1], H * W).to(torch.get_autocast_dtype('cuda'))
        else:
            x3 = x.view(B, x.shape[1], H * W)
        with torch.autocast("cuda", enabled=False):
            a3 = None
            if addend is not None:
                a3 = addend.contiguous().view(B, self.out_channels, H * W).to(x3.dtype)
            if ops.conv1x1_x3_supported(x3) and (H * W) % 4 == 0 and os.environ.get("MK_CONV_FP32", "x3") == "x3":
                y = _X3Conv.apply(x3, self.weight2d(), a3)
            else:
                y = _PointwiseConv.apply(x3, self.weight2d(), a3)
        if self.bias is not None and add_bias:
            y = y + self.bias.to(y.dtype).view(1, -1, 1)
        return y.view(B, self.out_channels, H, W)


def _x3_inference_conv(m, x, gelu):
    """fp32 field, grad mode off: ``act(conv(x) + bias)`` as ONE launch of the bf16x3 engine (bias and exact GELU in the epilogue,
    ``mk_conv1x1_x3_bias_act``), or None where that path does not apply."""
    from . import ops
    if torch.is_grad_enabled() or x.dim() != 4 or not x.is_cuda or not x.is_contiguous() or x.dtype != torch.float32:
        return None
    if torch.is_autocast_enabled() or os.environ.get("MK_CONV_FP32", "x3") != "x3":
        return None
    B, H, W = x.shape[0], x.shape[2], x.shape[3]
    x3 = x.view(B, x.shape[1], H * W)
    if not ops.conv1x1_x3_supported(x3) or (H * W) % 4 != 0 or m.weight.dtype != torch.float32:
        return None
    y = ops.conv1x1_x3(m.weight2d().detach(), x3, bias=None if m.bias is None else m.bias.detach(), gelu=gelu)
    return y.view(B, m.out_channels, H, W)


def _is_exact_gelu(m):
    return isinstance(m, nn.GELU) and getattr(m, "approximate", "none") == "none"


def _is_noop(m):
    return isinstance(m, nn.Identity) or (isinstance(m, (nn.Dropout, nn.Dropout2d)) and (m.p == 0.0 or not m.training))


def _mlp_pattern(mods):
    """``[Conv1x1, exact GELU, no-ops..., Conv1x1, no-ops...]`` -> (fc1, fc2), else None."""
    convs = [m for m in mods if isinstance(m, Conv1x1)]
    if len(convs) != 2 or not isinstance(mods[0], Conv1x1) or len(mods) < 3 or not _is_exact_gelu(mods[1]):
        return None
    rest = [m for m in mods[2:] if not _is_noop(m)]
    if len(rest) != 1 or rest[0] is not convs[1]:
        return None
    return convs[0], convs[1]


def run_pointwise_chain(mods, x, skip_last_bias=False, want_row_sums=False):
    """Evaluate an ``nn.Sequential`` of 1x1 convs / activations / identities.

    The two-convolution pattern of ``MLP`` and ``EncoderDecoder`` (conv + bias, exact GELU, conv) runs as one fused
    autograd node on the pixel-column engine when the field is bf16 (or autocast to it).  Otherwise every
    ``Conv1x1(bias) -> GELU`` pair is one GEMM plus one HIP bias+GELU pass.

    ``skip_last_bias``: the caller feeds the result straight into an instance norm without running statistics, which
    removes any per-channel constant -- the last conv's bias add is then a no-op on the output and its gradient is
    exactly zero, so the add is skipped (the parameter still receives that zero gradient).

    ``want_row_sums``: return ``(y, sums)`` with ``sums`` the fp64 ``[B * C, 2]`` per-row (sum, sum of squares) of ``y``
    when the engine produced them alongside (else ``None``) -- the statistics of the instance norm that follows.
    """
    from . import ops
    mods = list(mods)
    pat = _mlp_pattern(mods)
    if pat is not None and x.dim() == 4:
        fc1, fc2 = pat
        x3 = _engine_field(x)
        if (x3 is not None and ops.pce_supported_train(fc1.out_channels, fc1.in_channels)
                and ops.pce_supported_train(fc2.out_channels, fc2.in_channels)):
            fused = _mlp_fused_ok(fc1, fc2)
            with torch.autocast("cuda", enabled=False):
                if fused:
                    y, sums = _PceMLPFused.apply(x3, fc1.weight2d(), fc1.bias, fc2.weight2d(), fc2.bias, not skip_last_bias, want_row_sums)
                else:       # inference (grad mode off): the pre-activation is not kept -- one 768-row store less per MLP
                    y, sums = _PceMLP.apply(x3, fc1.weight2d(), fc1.bias, fc2.weight2d(), fc2.bias, not skip_last_bias, want_row_sums,
                                            torch.is_grad_enabled())
            y = y.view(x.shape[0], fc2.out_channels, x.shape[2], x.shape[3])
            return (y, sums if sums.numel() else None) if want_row_sums else y
    last_conv = max((j for j, m in enumerate(mods) if isinstance(m, Conv1x1)), default=-1)
    tail_is_noop = all(_is_noop(m) for m in mods[last_conv + 1:])
    i = 0
    while i < len(mods):
        m = mods[i]
        if skip_last_bias and tail_is_noop and i == last_conv and m.bias is not None:
            x = _ZeroGradBias.apply(m(x, add_bias=False), m.bias)
            i += 1
            continue
        if isinstance(m, Conv1x1) and i + 1 < len(mods) and _is_exact_gelu(mods[i + 1]) and x.is_cuda:
            y = _x3_inference_conv(m, x, True)            # fp32 inference: conv + bias + GELU in one launch
            if y is not None:
                x = y
                i += 2
                continue
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
    return (x, None) if want_row_sums else x


class _ZeroGradBias(torch.autograd.Function):
    """Identity on ``y`` that keeps a skipped bias in the graph: its gradient in front of an instance norm is zero."""

    @staticmethod
    def forward(ctx, y, bias):
        ctx.meta = (bias.dtype, tuple(bias.shape))
        return y.view_as(y)

    @staticmethod
    def backward(ctx, gy):
        return gy, torch.zeros(ctx.meta[1], dtype=ctx.meta[0], device=gy.device)


class InstanceNorm2d(nn.InstanceNorm2d):
    """``nn.InstanceNorm2d(affine=True, track_running_stats=False)`` as two HIP streaming passes
    (row sums, apply), optionally with the block's GELU fused into the apply pass."""

    def forward(self, x, fuse_gelu=False, row_sums=None):
        from . import ops
        if self.track_running_stats or not ops.pointwise_supported(x):
            y = super().forward(x)
            return F.gelu(y) if fuse_gelu else y
        return ops.instance_norm(x, self.weight, self.bias, self.eps, fuse_gelu, row_sums=row_sums)


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

    def _run(self, x, skip_last_bias=False, want_row_sums=False):
        return run_pointwise_chain(self.fwd, x, skip_last_bias, want_row_sums)

    def checkpoint_forward(self, x, skip_last_bias=False):
        return checkpoint(self._run, x, skip_last_bias, use_reentrant=False)

    def forward(self, x, skip_last_bias=False, want_row_sums=False):
        """``want_row_sums``: return ``(y, sums)`` -- see ``run_pointwise_chain``."""
        if self.checkpointing >= 2:
            y = self.checkpoint_forward(x, skip_last_bias)
            return (y, None) if want_row_sums else y
        return self._run(x, skip_last_bias, want_row_sums)


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
