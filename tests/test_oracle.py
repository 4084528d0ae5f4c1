"""CPU tests: pin the oracle against golden vectors and independent identities.

The reference ships no SHT vectors (parity unpinned there); the oracle is pinned
against scipy, analytic fields and the reference's own importable torch modules
(fixtures made by tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import sht, spectral, dist


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_quadrature_weights():
    for grid in ("equiangular", "legendre-gauss"):
        for n in (2, 3, 33, 240, 721):
            tq, w = sht.quadrature(grid, n)
            assert abs(w.sum() - 2.0) < 1e-12
            assert np.all(np.diff(tq) > 0) and tq[0] >= 0 and tq[-1] <= np.pi + 1e-12
    # exactness: CC with n nodes integrates polynomials of degree < n exactly
    x, w = sht.clenshaw_curtiss_weights(33)
    for d in range(0, 32, 2):
        assert abs((w * x**d).sum() - 2.0 / (d + 1)) < 1e-12
    x, w = sht.legendre_gauss_weights(12)
    for d in range(0, 24, 2):
        assert abs((w * x**d).sum() - 2.0 / (d + 1)) < 1e-12


def test_legendre_table_vs_scipy(golden_dir):
    g = _load(golden_dir, "sht_legendre_scipy_33.npz")
    tq, _ = sht.quadrature("equiangular", 33)
    assert np.allclose(tq, g["theta"], atol=1e-14)
    tab = sht.precompute_legpoly(16, 16, tq)
    assert np.abs(tab - g["table"]).max() < 1e-13


def test_analytic_fields(golden_dir):
    g = _load(golden_dir, "sht_analytic_33x64.npz")
    f = sht.RealSHT(33, 64, lmax=8, mmax=9, grid="equiangular", dtype=np.float64)
    c = f(g["fields"])
    assert np.abs(c - g["coeffs"]).max() < 1e-12


@pytest.mark.parametrize("nlat,nlon,lmax,mmax,grid", [
    (33, 64, 16, 17, "equiangular"), (91, 180, 30, 31, "equiangular"),
    (32, 64, 32, 33, "legendre-gauss"), (240, 480, 240, 241, "legendre-gauss"),
])
def test_roundtrip_and_adjoint(nlat, nlon, lmax, mmax, grid):
    rng = np.random.default_rng(333)
    f = sht.RealSHT(nlat, nlon, lmax, mmax, grid, dtype=np.float64)
    fi = sht.InverseRealSHT(nlat, nlon, lmax, mmax, grid, dtype=np.float64)
    c = np.zeros((2, lmax, mmax), dtype=np.complex128)
    for m in range(min(mmax, lmax)):
        c[:, m:, m] = rng.standard_normal((2, lmax - m)) + 1j * (m > 0) * rng.standard_normal((2, lmax - m))
    c2 = f(fi(c))
    assert np.linalg.norm(c2 - c) / np.linalg.norm(c) < 1e-11
    # fp32 variant meets the 1e-5 budget with margin
    f32 = sht.RealSHT(nlat, nlon, lmax, mmax, grid)
    fi32 = sht.InverseRealSHT(nlat, nlon, lmax, mmax, grid)
    c3 = f32(fi32(c.astype(np.complex64)))
    assert np.linalg.norm(c3 - c) / np.linalg.norm(c) < 5e-6


def test_torch_sht_matches_numpy_and_grad():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 33, 64)).astype(np.float32)
    fn, ft = sht.RealSHT(33, 64, 16, 17, "equiangular"), spectral.TorchRealSHT(33, 64, 16, 17, "equiangular")
    c_np = fn(x)
    xt = torch.from_numpy(x).requires_grad_(True)
    c_t = ft(xt)
    assert np.linalg.norm(c_t.detach().numpy() - c_np) / np.linalg.norm(c_np) < 1e-6
    it = spectral.TorchInverseRealSHT(33, 64, 16, 17, "equiangular")
    y = it(c_t)
    assert np.linalg.norm(y.detach().numpy() - sht.InverseRealSHT(33, 64, 16, 17, "equiangular")(c_np)) < 1e-4
    # adjointness of the real-linear maps under the torch gradient convention
    gy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, xt, gy)
    x2 = torch.randn_like(xt)
    lhs = (it(ft(x2)) * gy).sum()
    rhs = (x2 * gx).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4


def test_contractions_vs_reference(golden_dir):
    g = _load(golden_dir, "ref_contractions.npz")
    x = torch.from_numpy(g["x"])
    for name, fn in (("diagonal", spectral.contract_diagonal), ("dhconv", spectral.contract_dhconv)):
        y = fn(x, torch.from_numpy(g["w_" + name]))
        assert torch.allclose(y, torch.from_numpy(g["y_" + name]), rtol=1e-5, atol=1e-5)
    xr = torch.view_as_real(x).contiguous()
    for name, fn in (("diagonal_real", spectral.contract_diagonal_real), ("dhconv_real", spectral.contract_dhconv_real)):
        y = fn(xr, torch.from_numpy(g["w_" + name]))
        assert torch.allclose(y, torch.from_numpy(g["y_" + name]), rtol=1e-5, atol=1e-5)
    # the reference's separable einsums raise (output index 'o' has no source); so does the oracle
    with pytest.raises(RuntimeError):
        spectral.contract_sep_dhconv(x, torch.randn(6, 7, dtype=torch.complex64))


def test_layers_vs_reference(golden_dir):
    g = _load(golden_dir, "ref_layers.npz")
    mlp = spectral.MLP(6, 12, act_layer=torch.nn.GELU, gain=0.5)
    mlp.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("mlp.")}, strict=True)
    assert torch.allclose(mlp(torch.from_numpy(g["xm"])), torch.from_numpy(g["ym"]), rtol=1e-5, atol=1e-6)
    enc = spectral.EncoderDecoder(1, 4, 6, 6, torch.nn.GELU)
    enc.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("enc.")}, strict=True)
    assert torch.allclose(enc(torch.from_numpy(g["xe"])), torch.from_numpy(g["ye"]), rtol=1e-5, atol=1e-6)


def test_sfno_state_dict_keys_and_shapes():
    net = spectral.SphericalFourierNeuralOperatorNet(inp_shape=(33, 64), out_shape=(33, 64), scale_factor=2,
                                                     inp_chans=4, out_chans=2, embed_dim=8, num_layers=2)
    keys = set(net.state_dict().keys())
    want = {"encoder.fwd.0.weight", "encoder.fwd.0.bias", "encoder.fwd.2.weight", "decoder.fwd.0.weight",
            "decoder.fwd.0.bias", "decoder.fwd.2.weight", "residual_transform.weight"}
    for i in range(2):
        want |= {f"blocks.{i}.filter.filter.weight", f"blocks.{i}.norm0.weight", f"blocks.{i}.norm0.bias",
                 f"blocks.{i}.norm1.weight", f"blocks.{i}.norm1.bias", f"blocks.{i}.mlp.fwd.0.weight",
                 f"blocks.{i}.mlp.fwd.0.bias", f"blocks.{i}.mlp.fwd.3.weight", f"blocks.{i}.mlp.fwd.3.bias",
                 f"blocks.{i}.outer_skip.weight"}
    assert keys == want
    assert net.state_dict()["blocks.0.filter.filter.weight"].shape == (8, 8, 16)
    x = torch.randn(2, 4, 33, 64, requires_grad=True)
    y = net(x)
    assert y.shape == (2, 2, 33, 64)
    y.sum().backward()
    assert x.grad is not None and x.grad.shape == x.shape


def test_split_shapes():
    assert dist.compute_split_shapes(721, 8) == [91] * 7 + [84]
    assert dist.compute_split_shapes(721, 4) == [181, 181, 181, 178]
    assert dist.compute_split_shapes(240, 8) == [30] * 8
    assert dist.compute_split_shapes(384, 8) == [48] * 8
    assert dist.compute_split_shapes(9, 4) == [2, 2, 2, 3]   # ceil rule would leave 0 for the last shard
    assert sum(dist.compute_split_shapes(241, 7)) == 241
