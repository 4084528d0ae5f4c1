"""Oracle: spectral convolution, FNO block and SFNO network (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Plain restatement, module by module, of the reference's hot path:

* ``contract_*``            <- makani/models/common/contractions.py:121-178
* ``TorchRealSHT`` etc.     <- torch-harmonics RealSHT / InverseRealSHT (see oracle/sht.py)
* ``SpectralConv``          <- makani/models/common/spectral_convolution.py:43-148
* ``MLP`` / ``EncoderDecoder`` <- makani/models/common/layers.py:86-216
* ``FourierNeuralOperatorBlock`` <- makani/models/networks/sfnonet.py:119-267
* ``SphericalFourierNeuralOperatorNet`` <- makani/models/networks/sfnonet.py:270-640

State-dict keys equal the reference's so weights can be
copied verbatim between the oracle and the HIP-backed modules in tests.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import sht as _sht


# ----------------------------------------------------------------------------
# contractions (contractions.py:121-178)
# ----------------------------------------------------------------------------
def contract_diagonal(a, b):
    return torch.einsum("bixy,ioxy->boxy", a, b)


def contract_dhconv(a, b):
    return torch.einsum("bixy,iox->boxy", a, b)


def contract_sep_diagonal(a, b):
    return torch.einsum("bixy,ixy->boxy", a, b)


def contract_sep_dhconv(a, b):
    return torch.einsum("bixy,ix->boxy", a, b)


def contract_diagonal_real(a, b):
    return torch.einsum("bixys,ioxy->boxys", a, b).contiguous()


def contract_dhconv_real(a, b):
    return torch.einsum("bixys,iox->boxys", a, b).contiguous()


def contract_sep_diagonal_real(a, b):
    return torch.einsum("bixys,ixy->boxys", a, b).contiguous()


def contract_sep_dhconv_real(a, b):
    return torch.einsum("bixys,ix->boxys", a, b).contiguous()


def get_contract_fun(separable=False, operator_type="diagonal", complex=True):
    """factorizations.py:167-200 (dense path)."""
    table = {
        (True, "diagonal", True): contract_sep_diagonal,
        (True, "diagonal", False): contract_sep_diagonal_real,
        (True, "dhconv", True): contract_sep_dhconv,
        (True, "dhconv", False): contract_sep_dhconv_real,
        (False, "diagonal", True): contract_diagonal,
        (False, "diagonal", False): contract_diagonal_real,
        (False, "dhconv", True): contract_dhconv,
        (False, "dhconv", False): contract_dhconv_real,
    }
    key = (bool(separable), operator_type, bool(complex))
    if key not in table:
        raise ValueError(f"Unkonw operator type {operator_type}")
    return table[key]


# ----------------------------------------------------------------------------
# differentiable SHT on torch CPU tensors
# ----------------------------------------------------------------------------
class TorchRealSHT(nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss"):
        super().__init__()
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        tq, w = _sht.quadrature(grid, nlat)
        pct = _sht.precompute_legpoly(self.mmax, self.lmax, tq)
        self.register_buffer("weights", torch.from_numpy(pct * w[None, None, :]).float(), persistent=False)

    def forward(self, x):
        x = 2.0 * math.pi * torch.fft.rfft(x, dim=-1, norm="forward")
        x = torch.view_as_real(x[..., : self.mmax])
        w = self.weights.to(x.dtype)
        out = torch.einsum("...kmr,mlk->...lmr", x, w).contiguous()
        return torch.view_as_complex(out)


class TorchInverseRealSHT(nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss"):
        super().__init__()
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        tq, _ = _sht.quadrature(grid, nlat)
        pct = _sht.precompute_legpoly(self.mmax, self.lmax, tq)
        self.register_buffer("pct", torch.from_numpy(pct).float(), persistent=False)

    def forward(self, x):
        x = torch.view_as_real(x)
        p = self.pct.to(x.dtype)
        xs = torch.einsum("...lmr,mlk->...kmr", x, p).contiguous()
        x = torch.view_as_complex(xs)
        return torch.fft.irfft(x, n=self.nlon, dim=-1, norm="forward")


# ----------------------------------------------------------------------------
# SpectralConv (spectral_convolution.py:43-148)
# ----------------------------------------------------------------------------
class SpectralConv(nn.Module):
    def __init__(self, forward_transform, inverse_transform, in_channels, out_channels,
                 operator_type="diagonal", separable=False, bias=False, gain=1.0):
        super().__init__()
        self.forward_transform = forward_transform
        self.inverse_transform = inverse_transform
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes_lat = inverse_transform.lmax
        self.modes_lon = inverse_transform.mmax
        self.scale_residual = (forward_transform.nlat != inverse_transform.nlat) or (forward_transform.nlon != inverse_transform.nlon)
        if hasattr(forward_transform, "grid"):
            self.scale_residual = self.scale_residual or (forward_transform.grid != inverse_transform.grid)
        self.operator_type, self.separable = operator_type, separable

        weight_shape = [in_channels]
        if not separable:
            weight_shape += [out_channels]
        if operator_type == "diagonal":
            weight_shape += [self.modes_lat, self.modes_lon]
        elif operator_type == "dhconv":
            weight_shape += [self.modes_lat]
        else:
            raise ValueError(f"Unsupported operator type f{operator_type}")

        scale = math.sqrt(gain / in_channels) * torch.ones(self.modes_lat, dtype=torch.complex64)
        scale[0] *= math.sqrt(2.0)
        self.weight = nn.Parameter(scale * torch.randn(*weight_shape, dtype=torch.complex64))
        self._contract = get_contract_fun(separable=separable, operator_type=operator_type, complex=True)
        if bias == "constant":
            self.bias = nn.Parameter(torch.zeros(1, out_channels, 1, 1))
        elif bias == "position":
            self.bias = nn.Parameter(torch.zeros(1, out_channels, inverse_transform.nlat, inverse_transform.nlon))

    def forward(self, x):
        dtype = x.dtype
        residual = x
        x = x.float()
        x = self.forward_transform(x).contiguous()
        if self.scale_residual:
            residual = self.inverse_transform(x).to(dtype)
        x = self._contract(x, self.weight).contiguous()
        x = self.inverse_transform(x)
        if hasattr(self, "bias"):
            x = x + self.bias
        return x.to(dtype), residual


# ----------------------------------------------------------------------------
# glue layers (layers.py:86-216)
# ----------------------------------------------------------------------------
class EncoderDecoder(nn.Module):
    def __init__(self, num_layers, input_dim, output_dim, hidden_dim, act_layer, gain=1.0):
        super().__init__()
        mods, cur = [], input_dim
        for _ in range(num_layers):
            conv = nn.Conv2d(cur, hidden_dim, 1, bias=True)
            nn.init.normal_(conv.weight, mean=0.0, std=math.sqrt(2.0 / cur))
            nn.init.constant_(conv.bias, 0.0)
            mods += [conv, act_layer()]
            cur = hidden_dim
        conv = nn.Conv2d(cur, output_dim, 1, bias=False)
        nn.init.normal_(conv.weight, mean=0.0, std=math.sqrt(gain / cur))
        mods.append(conv)
        self.fwd = nn.Sequential(*mods)

    def forward(self, x):
        return self.fwd(x)


class MLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 output_bias=True, gain=1.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        fc1 = nn.Conv2d(in_features, hidden_features, 1, bias=True)
        nn.init.normal_(fc1.weight, mean=0.0, std=math.sqrt(2.0 / in_features))
        nn.init.constant_(fc1.bias, 0.0)
        fc2 = nn.Conv2d(hidden_features, out_features, 1, bias=output_bias)
        nn.init.normal_(fc2.weight, mean=0.0, std=math.sqrt(gain / hidden_features))
        if fc2.bias is not None:
            nn.init.constant_(fc2.bias, 0.0)
        # indices 2 and 4 are the (identity) dropouts of layers.py:206
        self.fwd = nn.Sequential(fc1, act_layer(), nn.Identity(), fc2, nn.Identity())

    def forward(self, x):
        return self.fwd(x)


class SpectralFilterLayer(nn.Module):
    """sfnonet.py:51-116, linear / non-factorized branch."""

    def __init__(self, forward_transform, inverse_transform, embed_dim, operator_type="diagonal",
                 separable=False, bias=False, gain=1.0):
        super().__init__()
        self.filter = SpectralConv(forward_transform, inverse_transform, embed_dim, embed_dim,
                                   operator_type=operator_type, separable=separable, bias=bias, gain=gain)

    def forward(self, x):
        return self.filter(x)


class FourierNeuralOperatorBlock(nn.Module):
    """sfnonet.py:119-267."""

    def __init__(self, forward_transform, inverse_transform, embed_dim, operator_type="diagonal",
                 mlp_ratio=2.0, act_layer=nn.GELU, norm_layer=(nn.Identity, nn.Identity),
                 separable=False, inner_skip="linear", outer_skip=None, use_mlp=False,
                 bias=False, final_activation=False):
        super().__init__()
        self.norm0 = norm_layer[0]()
        gain_factor = 1.0 if act_layer == nn.Identity else 2.0
        if inner_skip == "linear":
            self.inner_skip = nn.Conv2d(embed_dim, embed_dim, 1, 1, bias=False)
            gain_factor /= 2.0
            nn.init.normal_(self.inner_skip.weight, std=math.sqrt(gain_factor / embed_dim))
        elif inner_skip == "identity":
            self.inner_skip = nn.Identity()
            gain_factor /= 2.0
        elif inner_skip != "none":
            raise ValueError(f"Unknown skip connection type {inner_skip}")
        self.filter = SpectralFilterLayer(forward_transform, inverse_transform, embed_dim,
                                          operator_type=operator_type, separable=separable,
                                          bias=bias, gain=gain_factor)
        self.act_layer0 = act_layer()
        self.norm1 = norm_layer[1]()
        gain_factor = 2.0 if (final_activation and act_layer != nn.Identity) else 1.0
        if outer_skip == "linear":
            self.outer_skip = nn.Conv2d(embed_dim, embed_dim, 1, 1, bias=False)
            gain_factor /= 2.0
            nn.init.normal_(self.outer_skip.weight, std=math.sqrt(gain_factor / embed_dim))
        elif outer_skip == "identity":
            self.outer_skip = nn.Identity()
            gain_factor /= 2.0
        elif outer_skip not in ("none", None):
            raise ValueError(f"Unknown skip connection type {outer_skip}")
        if use_mlp:
            self.mlp = MLP(embed_dim, int(embed_dim * mlp_ratio), act_layer=act_layer, gain=gain_factor)
        if final_activation:
            self.act_layer1 = act_layer()

    def forward(self, x):
        x, residual = self.filter(x)
        x = self.norm0(x)
        if hasattr(self, "inner_skip"):
            x = x + self.inner_skip(residual)
        if hasattr(self, "act_layer0"):
            x = self.act_layer0(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        x = self.norm1(x)
        if hasattr(self, "outer_skip"):
            x = x + self.outer_skip(residual)
        if hasattr(self, "act_layer1"):
            x = self.act_layer1(x)
        return x


class ChannelLayerNorm(nn.Module):
    """``DistributedLayerNorm`` of makani/mpu/layer_norm.py:117-155 (single rank): NCHW -> channels last, ``nn.LayerNorm``
    (attribute ``norm``, so the parameter names match), back."""

    def __init__(self, num_features, eps=1e-6):
        super().__init__()
        self.norm = nn.LayerNorm(num_features, eps=eps, elementwise_affine=True)

    def forward(self, x):
        return torch.transpose(self.norm(torch.transpose(x, 1, 3)), 1, 3).contiguous()


class SphericalFourierNeuralOperatorNet(nn.Module):
    """sfnonet.py:270-640 (spectral_transform="sht", linear filter, serial)."""

    def __init__(self, model_grid_type="equiangular", sht_grid_type="legendre-gauss",
                 operator_type="dhconv", inp_shape=(721, 1440), out_shape=(721, 1440), scale_factor=8,
                 inp_chans=2, out_chans=2, embed_dim=32, num_layers=4, use_mlp=True, mlp_ratio=2.0,
                 encoder_ratio=1, decoder_ratio=1, activation_function="gelu", encoder_layers=1,
                 normalization_layer="instance_norm", max_modes=None, hard_thresholding_fraction=1.0,
                 big_skip=True, separable=False, bias=False, repeat_layers=1, pos_embed="none", **kwargs):
        super().__init__()
        self.repeat_layers = repeat_layers
        self.inp_shape, self.out_shape = inp_shape, out_shape
        self.inp_chans, self.out_chans, self.embed_dim = inp_chans, out_chans, embed_dim
        self.big_skip = big_skip
        self.h = int(inp_shape[0] // scale_factor)
        self.w = int(inp_shape[1] // scale_factor)
        if max_modes is not None:
            modes_lat, modes_lon = max_modes
        else:
            modes_lat = int(self.h * hard_thresholding_fraction)
            modes_lon = int((self.w // 2 + 1) * hard_thresholding_fraction)
        self.trans_down = TorchRealSHT(*inp_shape, lmax=modes_lat, mmax=modes_lon, grid=model_grid_type)
        self.itrans_up = TorchInverseRealSHT(*out_shape, lmax=modes_lat, mmax=modes_lon, grid=model_grid_type)
        self.trans = TorchRealSHT(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type)
        self.itrans = TorchInverseRealSHT(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type)

        act = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU}.get(activation_function)
        if act is None:
            raise ValueError(f"Unknown activation function {activation_function}")
        self.encoder = EncoderDecoder(encoder_layers, inp_chans, embed_dim, int(encoder_ratio * embed_dim), act)

        if normalization_layer == "instance_norm":
            def norm():
                return nn.InstanceNorm2d(embed_dim, eps=1e-6, affine=True, track_running_stats=False)
        elif normalization_layer == "none":
            norm = nn.Identity
        elif normalization_layer == "layer_norm":
            # sfnonet.py:371-373 with mpu/layer_norm.py:117-155: nn.LayerNorm over the channel axis, per grid point
            def norm():
                return ChannelLayerNorm(embed_dim, eps=1e-6)
        else:
            raise NotImplementedError(normalization_layer)

        self.blocks = nn.ModuleList()
        for i in range(num_layers):
            first, last = i == 0, i == num_layers - 1
            self.blocks.append(FourierNeuralOperatorBlock(
                self.trans_down if first else self.trans,
                self.itrans_up if last else self.itrans,
                embed_dim, operator_type=operator_type, mlp_ratio=mlp_ratio, act_layer=act,
                norm_layer=(norm, norm), inner_skip="none", outer_skip="linear", use_mlp=use_mlp,
                separable=separable, bias=bias))
        self.decoder = EncoderDecoder(encoder_layers, embed_dim, out_chans, int(decoder_ratio * embed_dim), act,
                                      gain=0.5 if big_skip else 1.0)
        if big_skip:
            self.residual_transform = nn.Conv2d(inp_chans, out_chans, 1, bias=False)
            nn.init.normal_(self.residual_transform.weight, mean=0.0, std=math.sqrt(0.5 / inp_chans))
        # position embedding (sfnonet.py:469-501): a grid field, or (re, im) spectral coefficients synthesised each call
        if pos_embed == "direct":
            self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *inp_shape))
            self.pos_embed.type = "direct"
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
        elif pos_embed == "frequency":
            rc = nn.Parameter(torch.tril(torch.randn(1, embed_dim, modes_lat, modes_lon), diagonal=0))
            cc = nn.Parameter(torch.tril(torch.randn(1, embed_dim, modes_lat, modes_lon - 1), diagonal=-1))
            nn.init.trunc_normal_(rc, std=0.02)
            nn.init.trunc_normal_(cc, std=0.02)
            self.pos_embed = nn.ParameterList([rc, cc])
            self.pos_embed.type = "frequency"
        elif pos_embed not in ("none", "None", None):
            raise ValueError("Unknown position embedding type")

    def forward(self, x):
        if self.big_skip:
            if self.out_shape != self.inp_shape:
                residual = self.itrans_up(self.trans_down(x.float()).contiguous()).to(x.dtype)
            else:
                residual = x
        x = self.encoder(x)
        if hasattr(self, "pos_embed"):                      # sfnonet.py:606-618
            if self.pos_embed.type == "frequency":
                pe = torch.stack([self.pos_embed[0], nn.functional.pad(self.pos_embed[1], (1, 0), "constant", 0)], dim=-1)
                pe = self.itrans_up(torch.view_as_complex(pe))
            else:
                pe = self.pos_embed
            x = x + pe
        for _ in range(self.repeat_layers):                 # sfnonet.py:574-585
            for blk in self.blocks:
                x = blk(x)
        x = self.decoder(x)
        if self.big_skip:
            x = x + self.residual_transform(residual)
        return x
