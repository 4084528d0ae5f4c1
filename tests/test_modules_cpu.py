"""CPU tests of the host-side mirror: construction, state-dict parity with the reference's
key list, parameter annotations, weight storage layout.  No kernel launches."""
import math
import os

import numpy as np
import pytest
import torch

from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet, FourierNeuralOperatorNet
from makani_amd.sht import RealSHT, InverseRealSHT
from makani_amd.spectral_convolution import SpectralConv, FactorizedSpectralConv
from oracle import spectral as osp

KW = dict(inp_shape=(33, 64), out_shape=(33, 64), scale_factor=2, inp_chans=4, out_chans=2, embed_dim=8, num_layers=2)


def test_state_dict_parity_with_oracle():
    net = SphericalFourierNeuralOperatorNet(**KW, some_unknown_makani_param=1)
    ref = osp.SphericalFourierNeuralOperatorNet(**KW)
    sd, sdo = net.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(sdo.keys())
    for k in sd:
        assert sd[k].shape == sdo[k].shape and sd[k].dtype == sdo[k].dtype, k
    # tables / twiddles are non-persistent buffers (strict loading of reference checkpoints)
    assert not any("weights" in k or "pct" in k or "twiddles" in k for k in sd)
    net.load_state_dict(sdo, strict=True)
    for k in sd:
        assert torch.equal(net.state_dict()[k], sdo[k])


def test_dhconv_weight_is_stored_degree_major():
    f, i = RealSHT(33, 64, 16, 17, "equiangular"), InverseRealSHT(33, 64, 16, 17, "equiangular")
    conv = SpectralConv(f, i, 6, 5, operator_type="dhconv")
    w = conv.weight
    assert tuple(w.shape) == (6, 5, 16) and w.dtype == torch.complex64
    assert w.permute(2, 0, 1).is_contiguous()
    assert w.is_shared_mp == ["matmul", "w"] and w.sharded_dims_mp == [None, None, "h"]
    # survives a state-dict round trip and an optimizer step
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    conv.load_state_dict(sd)
    assert conv.weight.permute(2, 0, 1).is_contiguous()
    conv.weight.grad = torch.ones_like(conv.weight)
    torch.optim.Adam([conv.weight], lr=1e-3).step()
    assert conv.weight.permute(2, 0, 1).is_contiguous()
    assert conv.scale_residual is False
    d = SpectralConv(f, InverseRealSHT(33, 64, 16, 17, "legendre-gauss"), 6, 5, operator_type="dhconv")
    assert d.scale_residual is True


def test_annotations_and_structure():
    net = SphericalFourierNeuralOperatorNet(**KW)
    assert net.encoder.fwd[0].weight.is_shared_mp == ["spatial"]
    assert net.residual_transform.weight.is_shared_mp == ["spatial"]
    assert net.blocks[0].filter.filter.forward_transform is net.trans_down
    assert net.blocks[1].filter.filter.inverse_transform is net.itrans_up
    assert (net.trans.nlat, net.trans.nlon, net.trans.lmax, net.trans.mmax, net.trans.grid) == (16, 32, 16, 17, "legendre-gauss")
    assert net.inp_shape_loc == (33, 64) and (net.h_loc, net.w_loc) == (16, 32)
    big = dict(KW, inp_shape=(721, 1440), out_shape=(721, 1440), scale_factor=3, embed_dim=4, num_layers=1)
    # production geometry: 240 x 241 modes on a 240 x 480 Legendre-Gauss inner grid (sfnonet.py:325-326,516-520)
    n2 = SphericalFourierNeuralOperatorNet(**big)
    assert (n2.h, n2.w, n2.trans.lmax, n2.trans.mmax) == (240, 480, 240, 241)


def test_errors_mirror_reference():
    f, i = RealSHT(9, 16, 4, 5, "equiangular"), InverseRealSHT(9, 16, 4, 5, "equiangular")
    with pytest.raises(ValueError):
        SpectralConv(f, i, 2, 2, operator_type="nope")
    with pytest.raises(ValueError):
        SphericalFourierNeuralOperatorNet(**dict(KW, activation_function="tanh"))
    with pytest.raises(ValueError):
        RealSHT(9, 16, 4, 5, "lobatto")
    with pytest.raises(NotImplementedError):
        FactorizedSpectralConv(f, i, 2, 2, operator_type="dhconv", factorization="ComplexTucker")
    fc = FactorizedSpectralConv(f, i, 2, 3, operator_type="dhconv", factorization="ComplexDense")
    assert tuple(fc.weight.to_tensor().shape) == (2, 3, 4)
    # the spectral ops never run on the CPU: no silent fallback
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f(torch.zeros(1, 9, 16))


def test_fft_variant_constructs():
    # "diagonal" init broadcasts a length-lmax scale against the mmax axis (spectral_convolution.py:98-101):
    # like the reference it only constructs when modes_lat == modes_lon
    with pytest.raises(RuntimeError):
        FourierNeuralOperatorNet(**dict(KW, inp_shape=(32, 64), out_shape=(32, 64), operator_type="diagonal"))
    net = FourierNeuralOperatorNet(**dict(KW, inp_shape=(32, 64), out_shape=(32, 64), operator_type="diagonal",
                                          max_modes=(16, 16)))
    x = torch.randn(1, 4, 32, 64)
    assert net(x).shape == (1, 2, 32, 64)


# --------------------------------------------------------------------------- losses (utils/losses.py, utils/grids.py)
@pytest.mark.parametrize("rule", ["naive", "legendre-gauss", "clenshaw-curtiss"])
@pytest.mark.parametrize("crop", [None, ((20, 48), (5, 8))])
def test_grid_quadrature_against_oracle(rule, crop):
    from makani_amd.losses import GridQuadrature
    from oracle import losses as ol
    shape = (33, 64)
    cs, co = crop if crop else (None, (0, 0))
    for normalize, pole in ((False, 0), (True, 2)):
        q = GridQuadrature(rule, shape, crop_shape=cs, crop_offset=co, normalize=normalize, pole_mask=pole)
        want = ol.quad_weight(rule, shape, cs, co, normalize, pole)
        np.testing.assert_allclose(q.quad_weight[0, 0].double().numpy(), want, rtol=2e-6, atol=1e-12)
    if crop is None and rule != "naive":
        assert abs(float(GridQuadrature(rule, shape).quad_weight.double().sum()) - 4 * math.pi) < 1e-5    # area of the sphere


@pytest.mark.parametrize("p,absolute,squared", [(2, False, False), (2, True, True), (2, True, False), (1, False, False),
                                                (1, True, False), (2, False, True)])
def test_geometric_lp_loss_against_oracle(p, absolute, squared):
    from makani_amd.losses import GeometricLpLoss
    from oracle import losses as ol
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 5, 33, 64
    prd, tar = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    chw = torch.rand(1, C, generator=g)
    loss = GeometricLpLoss((H, W), (H, W), (0, 0), p=p, absolute=absolute, squared=squared, quadrature_rule="legendre-gauss")
    got = loss(prd, tar, chw)
    q = ol.quad_weight("legendre-gauss", (H, W), (H, W), (0, 0), normalize=True)
    want = ol.geometric_lp_loss(prd.numpy(), tar.numpy(), chw.numpy(), q, p=p, absolute=absolute, squared=squared)
    assert abs(float(got) - want) < 2e-6 * max(1.0, abs(want))


def test_loss_handler_parses_like_the_reference():
    from types import SimpleNamespace
    from makani_amd.losses import LossHandler
    from oracle import losses as ol
    H, W, C = 33, 64, 4
    base = dict(n_future=0, img_shape_x=H, img_shape_y=W, img_crop_shape_x=H, img_crop_shape_y=W, img_crop_offset_x=0,
                img_crop_offset_y=0, N_out_channels=C, channel_names=["u10m", "sst", "t2m", "z500"], channel_weights="auto",
                model_grid_type="equiangular")
    g = torch.Generator().manual_seed(1)
    prd, tar = torch.randn(2, C, H, W, generator=g), torch.randn(2, C, H, W, generator=g)
    for spec, (p, rule, absolute, squared, weighted) in {
            "l2": (2, "naive", False, False, False), "geometric l2": (2, "naive", False, False, False),
            "absolute squared geometric l2": (2, "naive", True, True, False), "weighted geometric l1": (1, "naive", False, False, True),
            "pole-masked absolute l2": (2, "naive", True, False, False)}.items():
        h = LossHandler(SimpleNamespace(loss=spec, **base))
        h.train()
        chw = np.ones(C)
        if weighted:
            chw[1] = 0.0                                          # "sst" is switched off (losses.py:63-65)
        chw = (chw / chw.sum()).reshape(1, C)
        q = ol.quad_weight(rule, (H, W), (H, W), (0, 0), normalize=True, pole_mask=1 if "pole-masked" in spec else 0)
        want = ol.geometric_lp_loss(prd.numpy(), tar.numpy(), chw, q, p=p, absolute=absolute, squared=squared)
        assert abs(float(h(prd, tar, None)) - want) < 2e-6 * max(1.0, abs(want)), spec
    with pytest.raises(ValueError):
        LossHandler(SimpleNamespace(loss="huber", **base))


# --------------------------------------------------------------------------- registration (pyproject.toml:106-112)
def test_entry_points_resolve():
    import importlib
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        eps = tomli.load(f)["project"]["entry-points"]["makani.models"]
    assert set(eps) == {"SFNO_MI355X", "FNO_MI355X"}
    for target in eps.values():
        mod, name = target.split(":")
        assert issubclass(getattr(importlib.import_module(mod), name), torch.nn.Module)


def test_file_path_registration_like_model_registry():
    """model_registry.py:63-79: 'path/to/file.py:Name' through spec_from_file_location, then instantiate."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("SphericalFourierNeuralOperatorNet", os.path.join(root, "sfno_mi355x.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    net = module.SphericalFourierNeuralOperatorNet(inp_shape=(17, 32), out_shape=(17, 32), scale_factor=2, inp_chans=2,
                                                   out_chans=2, embed_dim=4, num_layers=1)
    assert isinstance(net, torch.nn.Module) and len(list(net.parameters())) > 0
