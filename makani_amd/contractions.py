"""Spectral filter contractions -- the contraction seam of the hot path.

``get_contract_fun`` has the signature and dispatch of
``makani/models/common/factorizations.py:212-247`` (dense weights) and returns
``f(x, weight, separable=..., operator_type=...)`` operating on the public layout
``x [B, I, L, M]`` complex64, like ``makani/models/common/contractions.py:121-178``.

* ``dhconv`` complex (the SFNO filter, ``einsum("bixy,iox->boxy")``): HIP / fp32 MFMA
  (``mk_dhconv_*``), here wrapped with the layout conversion; ``SpectralConv`` uses the
  packed op directly and skips the conversions.
* ``diagonal`` complex (unusable at the north-star size): HIP streaming kernels
  (``mk_diag_*``) on the public layout.  The ``_real`` variants (real weight on (re, im) pairs): the same kernels with
  the weight promoted to complex.
* separable variants: the reference's einsums name an output index ``o`` no operand
  carries and raise ``RuntimeError`` (contractions.py:139-152,169-178); so do these.
"""
from functools import partial

import torch

from . import ops


def _contract_dhconv(ac, bc):
    """``einsum("bixy,iox->boxy")`` through the HIP dhconv kernel.

    Dense semantics (every (x, y) entry), as the reference einsum: the triangular skip of
    the kernel is disabled by launching with a degree offset of Y (all modes count as <= l).
    """
    B, I, X, Y = ac.shape
    O = bc.shape[1]
    xp = ops.spec_pack(ac.reshape(B * I, X, Y).contiguous(), Y, 0)
    yp = ops.dhconv(xp, bc, B, Y, 0)
    return ops.spec_unpack(yp, Y, 0).reshape(B, O, X, Y)


def _contract_diagonal(ac, bc):
    """``einsum("bixy,ioxy->boxy")`` (contractions.py:121-127): HIP streaming kernels for complex64 device operands."""
    if ac.is_cuda and ac.dtype == torch.complex64 and bc.dtype == torch.complex64:
        return ops.diag_contract(ac, bc)
    return torch.einsum("bixy,ioxy->boxy", ac, bc)


def _contract_sep_diagonal(ac, bc):
    return torch.einsum("bixy,ixy->boxy", ac, bc)


def _contract_sep_dhconv(ac, bc):
    return torch.einsum("bixy,ix->boxy", ac, bc)


def _as_hip_complex(a, b):
    """(complex view of the real-pair input, real weight promoted to complex) when the HIP kernels apply, else None."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[-1] == 2:
        return torch.view_as_complex(a.contiguous()), torch.complex(b, torch.zeros_like(b))
    return None


def _contract_diagonal_real(a, b):
    """``einsum("bixys,ioxy->boxys")`` (contractions.py:155-160): a real weight acting on (re, im) pairs is the complex
    contraction with a zero imaginary part -- on the device it runs on the same HIP kernels as the complex variant."""
    hc = _as_hip_complex(a, b)
    if hc is not None:
        return torch.view_as_real(_contract_diagonal(*hc)).contiguous()
    return torch.einsum("bixys,ioxy->boxys", a, b).contiguous()


def _contract_dhconv_real(a, b):
    """``einsum("bixys,iox->boxys")`` (contractions.py:163-168), see ``_contract_diagonal_real``."""
    hc = _as_hip_complex(a, b)
    if hc is not None:
        return torch.view_as_real(_contract_dhconv(*hc)).contiguous()
    return torch.einsum("bixys,iox->boxys", a, b).contiguous()


def _contract_sep_diagonal_real(a, b):
    return torch.einsum("bixys,ixy->boxys", a, b).contiguous()


def _contract_sep_dhconv_real(a, b):
    return torch.einsum("bixys,ix->boxys", a, b).contiguous()


def _contract_dense_pytorch(x, weight, separable=False, operator_type="diagonal", complex=True):
    x = x.contiguous()
    if separable:
        if operator_type == "diagonal":
            x = _contract_sep_diagonal(x, weight) if complex else _contract_sep_diagonal_real(x, weight)
        elif operator_type == "dhconv":
            x = _contract_sep_dhconv(x, weight) if complex else _contract_sep_dhconv_real(x, weight)
        else:
            raise ValueError(f"Unkonw operator type {operator_type}")
    else:
        if operator_type == "diagonal":
            x = _contract_diagonal(x, weight) if complex else _contract_diagonal_real(x, weight)
        elif operator_type == "dhconv":
            x = _contract_dhconv(x, weight) if complex else _contract_dhconv_real(x, weight)
        else:
            raise ValueError(f"Unkonw operator type {operator_type}")
    return x.contiguous()


def _contract_dense_reconstruct(x, weight, separable=False, operator_type="diagonal", complex=True):
    if not torch.is_tensor(weight):
        weight = weight.to_tensor()
    return _contract_dense_pytorch(x, weight, separable=separable, operator_type=operator_type, complex=complex)


def get_contract_fun(weight, implementation="reconstructed", separable=False, operator_type="diagonal", complex=True):
    if implementation == "reconstructed":
        return partial(_contract_dense_reconstruct, separable=separable, complex=complex, operator_type=operator_type)
    if implementation == "factorized":
        if torch.is_tensor(weight):
            return partial(_contract_dense_pytorch, separable=separable, complex=complex, operator_type=operator_type)
        if hasattr(weight, "to_tensor"):  # dense factorized weight: reconstruct, then contract
            return partial(_contract_dense_reconstruct, separable=separable, complex=complex, operator_type=operator_type)
        raise ValueError(f"Got unexpected weight type of class {weight.__class__.__name__}")
    raise ValueError(f'Got {implementation=}, expected "reconstructed" or "factorized"')
