"""ctypes binding of libmakani_amd.so -- the C ABI declared in include/makani_amd.h.

The library is the product; there is no fallback.  If it is missing (and cannot be
built because hipcc is absent) importing any compute entry point raises.
"""
import ctypes
import os

# torch must be loaded first: libmakani_amd.so needs libamdhip64.so.7 and has to bind to the SAME HIP
# runtime instance torch brought (its streams and allocations are what the kernels receive); loading
# the extension first would pull a second runtime from /opt/rocm into the process.
import torch  # noqa: F401

from . import build as _build

_LIB = None

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/makani_amd.h
SIGNATURES = {
    "mk_version": (_c_int, []),
    "mk_last_error": (ctypes.c_char_p, []),
    "mk_quadrature": (_c_int, [_c_int, _c_int, _vp, _vp]),
    "mk_legendre_kpad": (_c_int, [_c_int]),
    "mk_legendre_table": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_fft_twiddle_len": (_c_int, [_c_int]),
    "mk_fft_twiddles": (_c_int, [_c_int, _vp]),
    "mk_rfft": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _vp]),
    "mk_irfft": (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _vp]),
    "mk_legendre_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_legendre_inv": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_legendre_x3_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int]),
    "mk_legendre_x3_split": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_legendre_fwd_x3": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_legendre_inv_x3": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_rfft_ex": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _vp]),
    "mk_irfft_ex": (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _vp]),
    "mk_rfft_pm": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _c_int, _vp]),
    "mk_irfft_pm": (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _c_int, _vp]),
    "mk_irfft_sums": (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _c_int,
                               _c_int, _vp, _vp]),
    "mk_legendre_fwd_x3_ex": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_legendre_inv_x3_ex": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_dhconv_fwd": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_dhconv_dgrad": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_dhconv_wgrad": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_dhconv_fwd_x3": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_dhconv_dgrad_x3": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_dhconv_wgrad_x3": (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    "mk_diag_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_diag_dgrad": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_diag_wgrad": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_spec_pack": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "mk_spec_unpack": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "mk_bias_gelu_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_bias_gelu_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_instnorm_fwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _c_float,
                                 _c_int, _vp]),
    "mk_instnorm_fwd_ex": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong,
                                    ctypes.c_longlong, _c_float, _c_int, _c_int, _vp]),
    "mk_instnorm_bwd_ex": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong,
                                    ctypes.c_longlong, _c_int, _c_int, _vp]),
    "mk_instnorm_bwd_wb": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, ctypes.c_longlong, _c_int, _vp]),
    "mk_wmse_fwd": (_c_int, [_vp, _c_int, _vp, _vp, _vp, ctypes.c_longlong, _c_int, _c_int, _c_float, _vp]),
    "mk_wmse_bwd": (_c_int, [_vp, _c_int, _vp, _vp, _vp, _vp, ctypes.c_longlong, _c_int, _c_int, _c_float, _vp]),
    "mk_conv1x1_wgrad": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_conv1x1_x3": (_c_int, [_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _c_int, _c_int, ctypes.c_longlong,
                               _c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _c_int, _vp]),
    "mk_conv1x1_x3_bias_act": (_c_int, [_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _c_int, _c_int,
                                        ctypes.c_longlong, _c_int, ctypes.c_longlong, ctypes.c_longlong, _vp, _c_int, _vp]),
    "mk_conv1x1_wgrad_act": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _c_int, _vp]),
    "mk_pce_mlp_image_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "mk_pce_mlp_pack": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "mk_pce_mlp": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                            ctypes.c_longlong, _vp]),
    "mk_pce_mlp_debug_stamps": (_c_int, [_vp]),
    "mk_pce_image_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "mk_pce_pack": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    "mk_pce_pack_layout": (_c_int, [_c_int, _c_int, _vp]),
    "mk_pce_pack_batch": (_c_int, [_vp, _c_int, _vp, ctypes.c_longlong, _vp]),
    "mk_pce_gemm": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_pce_gemm_ex": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp]),
    "mk_instnorm_coeffs": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, ctypes.c_longlong, ctypes.c_float, _vp]),
    "mk_pce_debug_stamps": (_c_int, [_vp]),
    "mk_adam_step": (_c_int, [_vp, _vp, _vp, _vp, ctypes.c_longlong] + [_c_float] * 5 + [_c_int, _vp]),
    "mk_instnorm_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong,
                                 _c_int, _vp]),
}


def lib_path():
    return _build.LIB


def load():
    """Load (building first if the sources are newer) and return the ctypes handle."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    override = os.environ.get("MK_LIB_OVERRIDE")        # A/B builds of the kernels (tools only): load this file as it is
    if override:
        path = override
    elif _build.stale():
        # missing, or older than a source / header: rebuild when hipcc is here, never run a stale library silently
        if _build.have_hipcc():
            path = _build.build(verbose=False)
        elif not os.path.exists(path):
            raise RuntimeError(f"makani_amd: {path} is missing and hipcc is not available to build it")
        else:
            raise RuntimeError(f"makani_amd: {path} is older than its sources and hipcc is not available to rebuild it")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - environment problem, never silent
        raise RuntimeError(f"makani_amd: cannot load the HIP extension {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mk_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"makani_amd {what} failed (code {rc}): {msg}")
