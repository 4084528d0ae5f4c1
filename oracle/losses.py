"""CPU restatement (numpy, float64) of the reference's sphere-quadrature Lp losses.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Follows makani/utils/grids.py:63-115 (GridQuadrature) and makani/utils/losses.py:174-271 (GeometricLpLoss).  Parity
unpinned by reference fixtures: the reference's tests hold no loss vectors (tests/test_trainer.py only runs the loop);
the quadrature weights underneath are pinned against torch-harmonics values by tests/test_oracle.py.
"""
import numpy as np

from .sht import clenshaw_curtiss_weights, legendre_gauss_weights


def quad_weight(rule, img_shape, crop_shape=None, crop_offset=(0, 0), normalize=False, pole_mask=0):
    """grids.py:67-109 -> [H, W] float64."""
    nlat, nlon = img_shape
    if rule == "naive":                                          # grids.py:67-75
        jac = np.clip(np.sin(np.linspace(0.0, np.pi, nlat)), 0.0, None)
        q = np.tile(((2 * np.pi / nlon) * (np.pi / nlat) * jac)[:, None], (1, nlon))
        q = q * (4.0 * np.pi) / q.sum()
    elif rule == "clenshaw-curtiss":                             # grids.py:76-81
        q = np.tile((2 * np.pi / nlon) * clenshaw_curtiss_weights(nlat, -1, 1)[1][:, None], (1, nlon))
    elif rule == "legendre-gauss":                               # grids.py:82-87
        q = np.tile((2 * np.pi / nlon) * legendre_gauss_weights(nlat, -1, 1)[1][:, None], (1, nlon))
    else:
        raise ValueError(rule)
    if normalize:                                                # grids.py:92-93
        q = q / (4.0 * np.pi)
    if pole_mask:                                                # grids.py:96-98 (with the evident meaning of `sizes`)
        q[:pole_mask] = 0.0
        q[nlat - pole_mask:] = 0.0
    if crop_shape is not None:                                   # grids.py:101-102
        q = q[crop_offset[0]:crop_offset[0] + crop_shape[0], crop_offset[1]:crop_offset[1] + crop_shape[1]]
    return q


def geometric_lp_loss(prd, tar, chw, q, p=2, absolute=False, squared=False, size_average=False):
    """losses.py:213-271: prd, tar [B, C, H, W]; chw [1, C]; q [H, W] -> scalar."""
    prd, tar = np.asarray(prd, np.float64), np.asarray(tar, np.float64)
    diff = (np.abs(prd - tar) ** p * q).sum(axis=(-2, -1))       # losses.py:216 / 236
    if absolute:
        norms = diff if squared else diff ** (1.0 / p)           # losses.py:219-220
    else:
        tarn = (np.abs(tar) ** p * q).sum(axis=(-2, -1))         # losses.py:239
        frac = diff / tarn                                       # losses.py:243
        norms = frac if squared else frac ** (1.0 / p)           # losses.py:245-246
    out = np.asarray(chw, np.float64) * norms                    # losses.py:223 / 249
    return out.mean() if size_average else out.sum()             # losses.py:225-229


def geometric_h1_loss(coeffs_diff, coeffs_tar=None, mask=None, alpha=0.5, squared=False, size_average=False):
    """losses.py:306-362 on given spherical-harmonic coefficients [B, C, L, M] (complex) of the error (and of the target for
    the relative form): norm2[l] = |c[l,0]|^2 + 2 sum_{m>0} |c[l,m]|^2, L2 = sum_l norm2, H1 = sum_l l (l + 1) norm2."""
    def norms(c):
        c = np.asarray(c)
        a = np.abs(c) ** 2
        norm2 = a[..., 0] + 2.0 * a[..., 1:].sum(axis=-1)                     # losses.py:310-311
        l = np.arange(c.shape[-2], dtype=np.float64)
        n = c.shape[0]
        return norm2.reshape(n, -1).sum(axis=-1), (norm2 * (l * (l + 1))).reshape(n, -1).sum(axis=-1)   # 312-313

    def combine(l2, h1):
        return alpha * l2 + (1 - alpha) * h1 if squared else alpha * np.sqrt(l2) + (1 - alpha) * np.sqrt(h1)   # 315-318

    out = combine(*norms(coeffs_diff))
    if coeffs_tar is not None:
        out = out / combine(*norms(coeffs_tar))                                # 348
        if mask is not None:
            out = out * np.asarray(mask, np.float64)                           # 349-350
            if size_average:
                return out.sum() / np.asarray(mask, np.float64).sum()          # 356-357
    return out.mean() if size_average else out.sum()
