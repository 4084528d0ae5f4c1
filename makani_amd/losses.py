"""Training losses of the Makani trainer for the hot path: quadrature-weighted Lp norms on the sphere.

Interface and arithmetic of ``makani/utils/grids.py:63-115`` (``GridQuadrature``) and ``makani/utils/losses.py:33-271``
(``LossHandler``, ``GeometricLpLoss``): per-sample, per-channel integrals of ``|prd - tar| ** p`` with the latitude
quadrature of the model grid, relative or absolute, optionally squared, channel-weighted, summed.  Under spatial model
parallelism prediction and target are gathered over ``h`` then ``w`` before the integral, as the reference does
(``losses.py:149-157``).

The quadrature weights depend on the latitude only.  For the absolute squared L2 norm with uniform channel weights --
the loss the bench harness trains with -- the whole thing is one streaming HIP pass over prediction and target
(``ops.weighted_mse`` / ``mk_wmse_fwd``, ``mk_wmse_bwd``); every other combination runs on torch ops.
"""
import math

import numpy as np
import torch
from torch import nn

from . import comm
from .distributed import compute_split_shapes
from .mappings import gather_from_parallel_region


def _latitude_weights(rule, nlat):
    """Quadrature weights over cos(theta) in [-1, 1] for the rules of torch_harmonics.quadrature."""
    from . import ops
    if rule == "legendre-gauss":
        return torch.from_numpy(ops.quadrature("legendre-gauss", nlat)[1])
    if rule == "clenshaw-curtiss":
        return torch.from_numpy(ops.quadrature("equiangular", nlat)[1])
    raise ValueError(f"Unknown quadrature rule {rule}")


class GridQuadrature(nn.Module):
    """``sum(x * quad_weight, dim=(-2, -1))`` with the weights of grids.py:63-115.

    (``pole_mask > 0`` raises ``NameError`` in the reference, grids.py:98 uses an undefined ``sizes``; here it masks the
    first / last ``pole_mask`` latitude rows, which is what the line above it does for the northern rows.)"""

    def __init__(self, quadrature_rule, img_shape, crop_shape=None, crop_offset=(0, 0), normalize=False, pole_mask=None):
        super().__init__()
        nlat, nlon = img_shape
        if quadrature_rule == "naive":
            jacobian = torch.clamp(torch.sin(torch.linspace(0, torch.pi, nlat)), min=0.0)
            quad_weight = (2 * torch.pi / nlon) * (torch.pi / nlat) * jacobian.unsqueeze(1)
            quad_weight = quad_weight.tile(1, nlon)
            quad_weight = quad_weight * (4.0 * torch.pi) / torch.sum(quad_weight)
        elif quadrature_rule in ("clenshaw-curtiss", "legendre-gauss"):
            w = _latitude_weights(quadrature_rule, nlat)
            quad_weight = ((2 * torch.pi / nlon) * w.unsqueeze(1)).tile(1, nlon)
        else:
            raise ValueError(f"Unknown quadrature rule {quadrature_rule}")
        if normalize:
            quad_weight = quad_weight / (4.0 * torch.pi)
        if (pole_mask is not None) and (pole_mask > 0):
            quad_weight[:pole_mask, :] = 0.0
            quad_weight[nlat - pole_mask:, :] = 0.0
        if crop_shape is not None:
            quad_weight = quad_weight[crop_offset[0]:crop_offset[0] + crop_shape[0], crop_offset[1]:crop_offset[1] + crop_shape[1]]
        quad_weight = quad_weight.contiguous()
        H, W = quad_weight.shape
        self.register_buffer("quad_weight", quad_weight.reshape(1, 1, H, W))

    def forward(self, x):
        return torch.sum(x * self.quad_weight, dim=(-2, -1))


class GeometricLpLoss(nn.Module):
    """Lp loss on the sphere, losses.py:174-271 (same constructor arguments, same ``forward(prd, tar, chw)``)."""

    def __init__(self, img_shape, crop_shape, crop_offset, p=2.0, size_average=False, reduction=True, absolute=False,
                 squared=False, pole_mask=0, jacobian="s2", quadrature_rule="naive"):
        super().__init__()
        self.p = p
        self.img_shape, self.crop_shape, self.crop_offset = img_shape, crop_shape, crop_offset
        self.reduction, self.size_average = reduction, size_average
        self.absolute, self.squared, self.pole_mask = absolute, squared, pole_mask
        self.quadrature = GridQuadrature(quadrature_rule, img_shape=img_shape, crop_shape=crop_shape, crop_offset=crop_offset,
                                         normalize=True, pole_mask=pole_mask)
        self.uniform_chw = None     # set by the owner of the channel weights when they are all equal (see _fused_abs_sq_l2)

    def _reduce(self, v):
        if not self.reduction:
            return v
        return torch.mean(v) if self.size_average else torch.sum(v)

    def _fused_abs_sq_l2(self, prd, tar, chw):
        """Absolute squared L2 with one weight for all channels: a single streaming pass (None if it does not apply)."""
        from . import ops
        if not (self.p == 2 and self.squared and self.reduction and not self.size_average and prd.is_cuda):
            return None
        if not (prd.dim() == 4 and prd.is_contiguous() and tar.is_contiguous() and tar.dtype == torch.float32
                and prd.dtype in (torch.float32, torch.bfloat16) and prd.shape[-1] % 8 == 0 and chw.numel() > 0):
            return None
        # one weight for all channels?  Known on the host by whoever built the weights (LossHandler sets ``uniform_chw``);
        # never read back from a device tensor here (that would be a synchronisation per step)
        value = self.uniform_chw
        if value is None and not chw.is_cuda:
            c0 = chw.detach().reshape(-1)
            if c0.numel() == prd.shape[1] and bool((c0 == c0[0]).all()):
                value = float(c0[0])
        if value is None:
            return None
        wrow = self.quadrature.quad_weight[0, 0, :, 0].contiguous()        # the weights do not vary along a latitude row
        return ops.weighted_mse(prd, tar, wrow, value)

    def abs(self, prd, tar, chw):
        fused = self._fused_abs_sq_l2(prd, tar, chw)
        if fused is not None:
            return fused
        num_examples = prd.size()[0]
        all_norms = self.quadrature(torch.abs(prd - tar) ** self.p).reshape(num_examples, -1)
        if not self.squared:
            all_norms = all_norms ** (1.0 / self.p)
        return self._reduce(chw * all_norms)

    def rel(self, prd, tar, chw):
        num_examples = prd.size()[0]
        diff_norms = self.quadrature(torch.abs(prd - tar) ** self.p).reshape(num_examples, -1)
        tar_norms = self.quadrature(torch.abs(tar) ** self.p).reshape(num_examples, -1)
        frac_norms = diff_norms / tar_norms
        if not self.squared:
            frac_norms = frac_norms ** (1.0 / self.p)
        return self._reduce(chw * frac_norms)

    def forward(self, prd, tar, chw):
        return self.abs(prd, tar, chw) if self.absolute else self.rel(prd, tar, chw)


class GeometricH1Loss(nn.Module):
    """Weighted H1 loss on the sphere, losses.py:275-370: L2 and l (l + 1)-weighted norms of the spherical-harmonic
    coefficients of the error (``alpha`` balances the two), relative to the target's or absolute.  The transform is this
    package's HIP ``RealSHT``; the degree sums are a handful of small torch reductions on its output.

    As in the reference the third argument of ``forward`` is a per-(sample, channel) mask of the relative form --
    ``LossHandler`` passes its channel weights there (losses.py:168 with 364-368), which weights the relative norms."""

    def __init__(self, img_shape, p=2.0, size_average=False, reduction=True, absolute=False, squared=False, alpha=0.5):
        super().__init__()
        from .sht import RealSHT
        self.reduction, self.size_average = reduction, size_average
        self.absolute, self.squared, self.alpha = absolute, squared, alpha
        self.sht = RealSHT(*img_shape, grid="equiangular").float()
        h1_weights = torch.arange(self.sht.lmax).float()
        self.register_buffer("h1_weights", h1_weights * (h1_weights + 1))

    def _norms(self, x):
        """(L2 norm squared, H1 seminorm squared) per sample: sums over channels, degrees and orders (m > 0 twice)."""
        c = torch.view_as_real(self.sht(x))
        c = c[..., 0] ** 2 + c[..., 1] ** 2
        norm2 = c[..., :, 0] + 2 * torch.sum(c[..., :, 1:], dim=-1)
        n = x.size()[0]
        return norm2.reshape(n, -1).sum(dim=-1), (norm2 * self.h1_weights).reshape(n, -1).sum(dim=-1)

    def _combine(self, l2, h1):
        if self.squared:
            return self.alpha * l2 + (1 - self.alpha) * h1
        return self.alpha * torch.sqrt(l2) + (1 - self.alpha) * torch.sqrt(h1)

    def abs(self, prd, tar):
        all_norms = self._combine(*self._norms(prd - tar))
        if self.reduction:
            return torch.mean(all_norms) if self.size_average else torch.sum(all_norms)
        return all_norms

    def rel(self, prd, tar, mask=None):
        retval = self._combine(*self._norms(prd - tar)) / self._combine(*self._norms(tar))
        if mask is not None:
            retval = retval * mask
        if self.reduction:
            if self.size_average:
                return torch.mean(retval) if mask is None else torch.sum(retval) / torch.sum(mask)
            return torch.sum(retval)
        return retval

    def forward(self, prd, tar, mask=None):
        return self.abs(prd, tar) if self.absolute else self.rel(prd, tar, mask)


class LossHandler(nn.Module):
    """losses.py:33-172 for the Lp family: parses ``params.loss`` ("l2", "geometric l2", "absolute squared geometric l2",
    "weighted ...", "pole-masked ...", "l1" ...), builds channel / multistep weights, gathers the spatial shards and
    calls the loss object.  ``params`` is the trainer's parameter object (attribute access)."""

    def __init__(self, params):
        super().__init__()
        self.rank = comm.get_rank("matmul")
        self.n_future = params.n_future
        self.img_shape = (params.img_shape_x, params.img_shape_y)
        self.crop_shape = (params.img_crop_shape_x, params.img_crop_shape_y)
        self.crop_offset = (params.img_crop_offset_x, params.img_crop_offset_y)
        self.loss_type = params.loss
        loss_type = set(self.loss_type.split())
        pole_mask = 1 if "pole-masked" in loss_type else 0
        if "weighted" in loss_type:
            if params.channel_weights == "auto":
                channel_weights = torch.ones(params.N_out_channels, dtype=torch.float32)
                for c, chn in enumerate(params.channel_names):
                    channel_weights[c] = 0.0 if chn in ["sst"] else 1.0
            else:
                channel_weights = torch.Tensor(params.channel_weights).float()
        else:
            channel_weights = torch.ones(params.N_out_channels, dtype=torch.float32)
        channel_weights = channel_weights.reshape(1, -1, 1, 1)
        channel_weights = channel_weights / torch.sum(channel_weights)
        absolute, squared = "absolute" in loss_type, "squared" in loss_type
        if "temp-std" in loss_type:
            eps = 1e-6
            global_stds = torch.from_numpy(np.load(params.global_stds_path)).reshape(1, -1, 1, 1)[:, params.out_channels]
            time_diff_stds = math.sqrt(params.dt) * torch.from_numpy(np.load(params.time_diff_stds_path)).reshape(1, -1, 1, 1)[:, params.out_channels]
            time_var_weights = global_stds / (time_diff_stds + eps)
            if squared:
                time_var_weights = time_var_weights ** 2
            channel_weights = channel_weights * time_var_weights
        self.register_buffer("channel_weights", channel_weights)
        rule = "legendre-gauss" if params.model_grid_type == "legendre_gauss" else "naive"
        common = dict(absolute=absolute, pole_mask=pole_mask)
        if "l2" in loss_type:
            if "geometric" in loss_type:
                self.loss_obj = GeometricLpLoss(self.img_shape, self.crop_shape, self.crop_offset, p=2, squared=squared,
                                                quadrature_rule=rule, **common)
            else:
                self.loss_obj = GeometricLpLoss(self.img_shape, self.crop_shape, self.crop_offset, p=2, jacobian="flat", **common)
        elif "l1" in loss_type:
            if "geometric" in loss_type:
                self.loss_obj = GeometricLpLoss(self.img_shape, self.crop_shape, self.crop_offset, p=1, quadrature_rule=rule, **common)
            else:
                self.loss_obj = GeometricLpLoss(self.img_shape, self.crop_shape, self.crop_offset, p=1, jacobian="flat", **common)
        # losses.py:129-130 tests the two-word string against the SET of words, so the reference never reaches this branch
        # (it raises "Unknown loss function"); the evident intent is kept: "geometric h1" selects the H1 loss
        elif "geometric h1" in self.loss_type:
            self.loss_obj = GeometricH1Loss(self.img_shape, absolute=absolute, squared=squared)
        else:
            raise ValueError(f"Unknown loss function: {self.loss_type}")
        multistep_weight = torch.ones(self.n_future + 1, dtype=torch.float32) / float(self.n_future + 1)
        self.register_buffer("multistep_weight", multistep_weight.reshape(-1, 1, 1, 1))
        # host-side knowledge for the fused pass: the value of the channel weights if they are all equal (training / eval)
        cw_train = (channel_weights * self.multistep_weight).reshape(-1)
        cw_eval = channel_weights.reshape(-1)
        self._uniform = {True: float(cw_train[0]) if bool((cw_train == cw_train[0]).all()) else None,
                         False: float(cw_eval[0]) if bool((cw_eval == cw_eval[0]).all()) else None}
        self.do_gather_input = comm.get_size("spatial") > 1
        if self.do_gather_input:
            self.gather_shapes_h = compute_split_shapes(self.crop_shape[0], comm.get_size("h"))
            self.gather_shapes_w = compute_split_shapes(self.crop_shape[1], comm.get_size("w"))

    def _gather_input(self, x):
        xh = gather_from_parallel_region(x, -2, self.gather_shapes_h, "h")
        return gather_from_parallel_region(xh, -1, self.gather_shapes_w, "w")

    def is_distributed(self):
        return False

    def forward(self, prd, tar, inp=None):
        if self.do_gather_input:
            prd, tar = self._gather_input(prd), self._gather_input(tar)
        chw = self.channel_weights
        chw = (chw * self.multistep_weight).reshape(1, -1) if self.training else chw.reshape(1, -1)
        if isinstance(self.loss_obj, GeometricLpLoss):
            self.loss_obj.uniform_chw = self._uniform[bool(self.training)]
        if isinstance(self.loss_obj, GeometricH1Loss):
            # the H1 loss reduces over channels itself ([B] norms): its ``mask`` is per SAMPLE, the per-channel weights of
            # the Lp family do not apply (losses.py:331-345); passing them broadcast only for B == 1 or B == C
            return self.loss_obj(prd, tar)
        return self.loss_obj(prd, tar, chw)
