"""Generate the committed golden fixtures under tests/golden/.

Run in the build container (the reference tree is mounted read-only there):

    python tests/golden/make_golden.py

Two families of vectors:

1. ``ref_*.npz`` -- inputs + outputs produced by IMPORTING the reference's own
   torch-only modules ``makani/models/common/contractions.py`` and ``layers.py``
   from /root/reference (package ``__init__`` files bypassed, since
   ``makani/__init__.py`` eagerly imports the trainer and its absent
   dependencies).  These pin the oracle's contraction / MLP / EncoderDecoder /
   RealFFT2 restatements to the reference itself.
2. ``sht_*.npz`` -- known answers from sources independent of the oracle: scipy's
   ``sph_harm_y`` (Legendre table), analytic fields.  The reference holds no SHT
   vectors (parity unpinned there, see oracle/__init__.py).

The fixtures are data only (inputs, expected outputs); no reference source text
is stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _load_reference_modules():
    """Import contractions.py and layers.py without running makani/__init__.py."""
    for name in ("makani", "makani.models", "makani.models.common"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    con = load("makani.models.common.contractions", "makani/models/common/contractions.py")
    lay = load("makani.models.common.layers", "makani/models/common/layers.py")
    return con, lay


def make_reference_vectors():
    con, lay = _load_reference_modules()
    g = torch.Generator().manual_seed(333)

    def crandn(*shape):
        return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))

    B, I, O, X, Y = 2, 6, 5, 7, 8
    x = crandn(B, I, X, Y)
    out = {"x": x.numpy()}
    # NOTE: the reference's separable einsums ("bixy,ixy->boxy", contractions.py:139-152,
    # 169-178) name an output index that no operand carries; torch raises RuntimeError
    # for them, so there is no reference output to record for the separable variants.
    w = {"diagonal": crandn(I, O, X, Y), "dhconv": crandn(I, O, X)}
    fns = {"diagonal": con._contract_diagonal, "dhconv": con._contract_dhconv}
    for name in ("_contract_sep_diagonal", "_contract_sep_dhconv"):
        try:
            getattr(con, name)(x, crandn(I, X, Y) if "diagonal" in name else crandn(I, X))
            raise SystemExit(f"{name} unexpectedly ran; add it to the fixtures")
        except RuntimeError:
            pass
    for k in w:
        out["w_" + k] = w[k].numpy()
        out["y_" + k] = fns[k](x, w[k]).numpy()
    # real-view variants (weights real, input as [..., 2])
    xr = torch.view_as_real(x).contiguous()
    wr = {"diagonal_real": torch.randn(I, O, X, Y, generator=g), "dhconv_real": torch.randn(I, O, X, generator=g)}
    fr = {"diagonal_real": con._contract_diagonal_real, "dhconv_real": con._contract_dhconv_real}
    for k in wr:
        out["w_" + k] = wr[k].numpy()
        out["y_" + k] = fr[k](xr, wr[k]).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_contractions.npz"), **out)

    # MLP / EncoderDecoder (layers.py:86-216): save weights so RNG does not matter
    torch.manual_seed(333)
    mlp = lay.MLP(in_features=6, hidden_features=12, act_layer=torch.nn.GELU, gain=0.5)
    enc = lay.EncoderDecoder(num_layers=1, input_dim=4, output_dim=6, hidden_dim=6, act_layer=torch.nn.GELU, gain=1.0)
    xm = torch.randn(2, 6, 5, 9, generator=g)
    xe = torch.randn(2, 4, 5, 9, generator=g)
    out = {"xm": xm.numpy(), "ym": mlp(xm).detach().numpy(), "xe": xe.numpy(), "ye": enc(xe).detach().numpy()}
    for k, v in mlp.state_dict().items():
        out["mlp." + k] = v.numpy()
    for k, v in enc.state_dict().items():
        out["enc." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_layers.npz"), **out)

    # RealFFT2 / InverseRealFFT2 (layers.py:219-287): the duck-typed transform seam
    f = lay.RealFFT2(16, 32, lmax=10, mmax=9)
    fi = lay.InverseRealFFT2(16, 32, lmax=10, mmax=9)
    xf = torch.randn(2, 3, 16, 32, generator=g)
    yf = f(xf)
    np.savez_compressed(os.path.join(HERE, "ref_fft2.npz"), x=xf.numpy(), y=yf.numpy(), xi=fi(yf).numpy())


def make_sht_vectors():
    from scipy.special import sph_harm_y

    nlat, lm = 33, 16
    theta = np.linspace(0.0, np.pi, nlat)  # equiangular colatitudes, north pole first
    tab = np.zeros((lm, lm, nlat))
    for m in range(lm):
        for l in range(m, lm):
            tab[m, l] = sph_harm_y(l, m, theta, 0.0).real
    np.savez_compressed(os.path.join(HERE, "sht_legendre_scipy_33.npz"), theta=theta, table=tab)

    # analytic fields on a 33 x 64 equiangular grid -> expected coefficients
    nlon = 64
    phi = 2 * np.pi * np.arange(nlon) / nlon
    T, P = np.meshgrid(theta, phi, indexing="ij")
    fields = np.stack([np.full_like(T, 1.5), np.cos(T), np.sin(T) * np.cos(P), np.sin(T) * np.sin(P),
                       3 * np.cos(T) ** 2 - 1])
    # (l, m, value): Y_00 = 1/sqrt(4pi); cos = sqrt(4pi/3) Y_10; sin cos(phi) = -sqrt(2pi/3) (Y_11 - Y_1-1) ...
    expect = np.zeros((5, 8, 9), dtype=np.complex128)
    expect[0, 0, 0] = 1.5 * np.sqrt(4 * np.pi)
    expect[1, 1, 0] = np.sqrt(4 * np.pi / 3)
    expect[2, 1, 1] = -np.sqrt(2 * np.pi / 3)
    expect[3, 1, 1] = 1j * np.sqrt(2 * np.pi / 3)
    expect[4, 2, 0] = 4 * np.sqrt(np.pi / 5)
    np.savez_compressed(os.path.join(HERE, "sht_analytic_33x64.npz"), fields=fields, coeffs=expect)


if __name__ == "__main__":
    make_reference_vectors()
    make_sht_vectors()
    print("golden fixtures written to", HERE)
