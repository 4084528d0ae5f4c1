"""Oracle: real spherical-harmonic transform (numpy, float64 tables).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Restates the published algorithm of torch-harmonics 0.6.x (dependency of the
reference, call sites ``makani/models/networks/sfnonet.py:524-539``,
``makani/models/common/spectral_convolution.py:131-141``).  Specification as
written down in SURVEY.md section 8(c):

* quadrature:  ``equiangular`` -> Clenshaw-Curtis on cos(theta), nodes
  ``cos(linspace(pi, 0, nlat))``; ``legendre-gauss`` -> ``leggauss(nlat)``.
* Legendre table ``P[m, l, k]`` by the float64 three-term recursion, orthonormal
  ("ortho"), Condon-Shortley phase, zero for l < m.
* forward:  ``X = 2*pi*rfft(x, norm="forward")[..., :mmax]``;
  ``c[..., l, m] = sum_k P[m, l, k] w_k X[..., k, m]``.
* inverse:  ``X[..., k, m] = sum_l P[m, l, k] c[..., l, m]``;
  ``x = irfft(X, n=nlon, norm="forward")``.
"""
import numpy as np


# ----------------------------------------------------------------------------
# quadrature rules
# ----------------------------------------------------------------------------
def legendre_gauss_weights(n, a=-1.0, b=1.0):
    """Gauss-Legendre nodes (ascending in cos(theta)) and weights on [a, b]."""
    xlg, wlg = np.polynomial.legendre.leggauss(n)
    xlg = (b - a) * 0.5 * xlg + (b + a) * 0.5
    wlg = wlg * (b - a) * 0.5
    return xlg, wlg


def clenshaw_curtiss_weights(n, a=-1.0, b=1.0):
    """Clenshaw-Curtis nodes ``cos(linspace(pi, 0, n))`` and weights on [a, b].

    Weights via the closed form  w_j = (c_j / n1) * (1 - sum_k b_k/(4k^2-1) cos(2 k j pi / n1)),
    (Waldvogel 2006 / Trefethen), evaluated directly in float64.  This is an
    independent formula from the FFT-based one torch-harmonics uses; both give
    the exact Clenshaw-Curtis rule.
    """
    assert n > 1
    tcc = np.cos(np.linspace(np.pi, 0, n))
    n1 = n - 1
    if n == 2:
        wcc = np.array([1.0, 1.0])
    else:
        j = np.arange(n)
        theta = np.pi * j / n1
        wcc = np.ones(n)
        for k in range(1, n1 // 2 + 1):
            bk = 1.0 if (2 * k == n1) else 2.0
            wcc -= bk / (4.0 * k * k - 1.0) * np.cos(2.0 * k * theta)
        c = np.full(n, 2.0)
        c[0] = c[-1] = 1.0
        wcc = c / n1 * wcc
    tcc = (b - a) * 0.5 * tcc + (b + a) * 0.5
    wcc = wcc * (b - a) * 0.5
    return tcc, wcc


def quadrature(grid, nlat):
    """Returns (colatitudes theta ascending from the north pole, weights)."""
    if grid == "legendre-gauss":
        cost, w = legendre_gauss_weights(nlat)
    elif grid == "equiangular":
        cost, w = clenshaw_curtiss_weights(nlat)
    else:
        raise ValueError(f"Unknown quadrature mode {grid}")
    # cost ascends from -1 (south) to 1 (north); torch-harmonics flips the
    # colatitudes so that row 0 is the north pole.  Both rules are symmetric, so
    # the weights need no flip.
    tq = np.flip(np.arccos(cost)).copy()
    return tq, w


# ----------------------------------------------------------------------------
# associated Legendre table
# ----------------------------------------------------------------------------
def precompute_legpoly(mmax, lmax, t, csphase=True):
    """Orthonormal associated Legendre functions P[m, l, k] at colatitudes t."""
    nmax = max(mmax, lmax)
    cost = np.cos(t)
    vdm = np.zeros((nmax, nmax, len(t)), dtype=np.float64)
    vdm[0, 0, :] = 1.0 / np.sqrt(4 * np.pi)
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * cost * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = np.sqrt((2 * l + 1) * (1 + cost) * (1 - cost) / 2 / l) * vdm[l - 1, l - 1, :]
    for l in range(2, nmax):
        m = np.arange(0, l - 1)[:, None]
        a = np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m))
        b = np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m))
        vdm[: l - 1, l, :] = cost[None, :] * a * vdm[: l - 1, l - 1, :] - b * vdm[: l - 1, l - 2, :]
    vdm = vdm[:mmax, :lmax]
    if csphase:
        vdm[1::2] *= -1
    return vdm


class RealSHT:
    """Forward transform: real [..., nlat, nlon] -> complex [..., lmax, mmax]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", dtype=np.float32):
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        tq, w = quadrature(grid, nlat)
        pct = precompute_legpoly(self.mmax, self.lmax, tq)
        self.dtype = np.dtype(dtype)
        # weights[m, l, k] = P[m, l, k] * w[k], cast like ``.float()`` at sfnonet.py:536
        self.weights = (pct * w[None, None, :]).astype(self.dtype)

    def __call__(self, x):
        x = np.asarray(x)
        cdt = np.complex64 if self.dtype == np.float32 else np.complex128
        X = (2.0 * np.pi) * np.fft.rfft(x.astype(self.dtype), axis=-1, norm="forward")
        X = X[..., : self.mmax].astype(cdt)
        re = np.einsum("...km,mlk->...lm", X.real.astype(self.dtype), self.weights, optimize=True)
        im = np.einsum("...km,mlk->...lm", X.imag.astype(self.dtype), self.weights, optimize=True)
        return (re + 1j * im).astype(cdt)


class InverseRealSHT:
    """Inverse transform: complex [..., lmax, mmax] -> real [..., nlat, nlon]."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", dtype=np.float32):
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        tq, _ = quadrature(grid, nlat)
        self.dtype = np.dtype(dtype)
        self.pct = precompute_legpoly(self.mmax, self.lmax, tq).astype(self.dtype)

    def __call__(self, c):
        c = np.asarray(c)
        cdt = np.complex64 if self.dtype == np.float32 else np.complex128
        re = np.einsum("...lm,mlk->...km", c.real.astype(self.dtype), self.pct, optimize=True)
        im = np.einsum("...lm,mlk->...km", c.imag.astype(self.dtype), self.pct, optimize=True)
        X = (re + 1j * im).astype(cdt)
        x = np.fft.irfft(X, n=self.nlon, axis=-1, norm="forward")
        return x.astype(self.dtype)
