"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares,
and its host-side float64 precompute agrees with the oracle.  No kernel launches here."""
import os
import re

import numpy as np
import pytest
import torch

from makani_amd import _lib, ops
from oracle import sht as osht

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "makani_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/makani_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mk_version() >= 100


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
@pytest.mark.parametrize("nlat", [2, 3, 33, 240, 721])
def test_quadrature_matches_oracle(grid, nlat):
    t, w = ops.quadrature(grid, nlat)
    to, wo = osht.quadrature(grid, nlat)
    assert np.abs(t - to).max() < 1e-13
    # Newton-on-P_n (C) vs numpy's eigenvalue leggauss: both float64, agree far below fp32 resolution
    assert np.abs(w - wo).max() < 1e-12 and np.abs(w / wo - 1).max() < 1e-8


@pytest.mark.parametrize("grid,nlat,lmax,mmax", [("equiangular", 33, 16, 17), ("legendre-gauss", 32, 32, 33),
                                                 ("equiangular", 721, 240, 241), ("legendre-gauss", 240, 240, 241)])
def test_legendre_table_matches_oracle(grid, nlat, lmax, mmax):
    tq, w = osht.quadrature(grid, nlat)
    ref = osht.precompute_legpoly(mmax, lmax, tq)
    for wq in (False, True):
        tab = ops.legendre_table(grid, nlat, lmax, mmax, wq).numpy()
        assert tab.shape == (mmax, lmax, ops.legendre_kpad(nlat))
        want = (ref * w[None, None, :] if wq else ref).astype(np.float32)
        got = tab[:, :, :nlat]
        # same float64 recursion rounded once to fp32: allow 1 ulp-scale differences from libm
        assert np.abs(got - want).max() <= 2e-7 * max(1.0, np.abs(want).max())
        assert np.all(tab[:, :, nlat:] == 0)
        for m in range(1, mmax):
            assert np.all(got[m, : min(m, lmax)] == 0)


def test_twiddles():
    for n in (16, 480, 1440):
        tw = ops.fft_twiddles(n).numpy().astype(np.float64)
        h = n // 2
        a = tw[: 2 * h].reshape(h, 2)
        b = tw[2 * h:].reshape(h + 1, 2)
        assert np.abs(a[:, 0] + 1j * a[:, 1] - np.exp(-2j * np.pi * np.arange(h) / h)).max() < 1e-7
        assert np.abs(b[:, 0] + 1j * b[:, 1] - np.exp(-2j * np.pi * np.arange(h + 1) / n)).max() < 1e-7


def test_errors_are_loud():
    lib = _lib.load()
    assert lib.mk_quadrature(7, 10, 0, 0) != 0
    assert b"unknown grid" in lib.mk_last_error()
    with pytest.raises(ValueError):
        ops.quadrature("lobatto", 8)
    # CPU tensors never reach a kernel, and there is no CPU fallback
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rfft_raw(torch.zeros(1, 4, 8), ops.fft_twiddles(8), 5, 1.0, 1.0, 1.0)
