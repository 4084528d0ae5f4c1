"""CPU tests of the host-side mirror: construction, state-dict parity with the reference's
key list, parameter annotations, weight storage layout.  No kernel launches."""
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
