"""ctypes binding of libbattgp.so (the C-ABI declared in include/battgp.h).

The library is the product: there is NO CPU or PyTorch fallback.  If the shared object is
missing or does not export the declared symbols, loading fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGP_EXPERIMENTAL_LIB=1 (builder sessions only: the A/B stage of tools/gpu_session.sh, the optional parity cases) selects the
# library built with -DBGP_EXPERIMENTAL (battgp_amd/build.py --experimental); same C-ABI, plus the untimed optional kernels
EXPERIMENTAL = os.environ.get("BGP_EXPERIMENTAL_LIB") == "1"
LIB_PATH = os.path.join(_HERE, "libbattgp_exp.so" if EXPERIMENTAL else "libbattgp.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
handle_p = C.c_void_p

# name -> (restype, argtypes); must list EVERY symbol of include/battgp.h
SIGNATURES = {
    "bgp_version": (C.c_int, []),
    "bgp_create": (C.c_int, [C.POINTER(handle_p), C.c_int]),
    "bgp_destroy": (None, [handle_p]),
    "bgp_trim": (C.c_int, [C.c_int]),
    "bgp_last_error": (C.c_char_p, [handle_p]),
    "bgp_set_kernel": (C.c_int, [handle_p, C.c_int, c_double_p, C.c_int]),
    "bgp_set_options": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_double, C.c_int]),
    "bgp_set_panel_scheme": (C.c_int, [handle_p, C.c_int]),
    "bgp_set_layout": (C.c_int, [handle_p, C.c_int64]),
    "bgp_get_layout": (C.c_int, [handle_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bgp_fit": (C.c_int, [handle_p, c_double_p, c_double_p, C.c_int64, C.c_int, c_double_p, c_double_p]),
    "bgp_fit_dev": (C.c_int, [handle_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, c_double_p, c_double_p]),
    "bgp_refit": (C.c_int, [handle_p, c_double_p, C.c_int, c_double_p, c_double_p]),
    "bgp_fit_predict": (
        C.c_int,
        [handle_p, c_double_p, c_double_p, C.c_int64, C.c_int, c_double_p, C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double],
    ),
    "bgp_fit_predict_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, c_double_p, c_double_p, C.c_void_p, C.c_void_p, C.c_double],
    ),
    "bgp_lml_grad": (C.c_int, [handle_p, c_double_p, C.c_int]),
    "bgp_set_keep_factor": (C.c_int, [handle_p, C.c_int]),
    "bgp_predict": (C.c_int, [handle_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_double]),
    "bgp_predict_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double]),
    "bgp_predict_cov": (C.c_int, [handle_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    "bgp_kernel_matrix": (C.c_int, [handle_p, c_double_p, C.c_int64, c_double_p, C.c_int64, C.c_int, c_double_p]),
    "bgp_get_alpha": (C.c_int, [handle_p, c_double_p]),
    "bgp_residuals": (C.c_int, [handle_p, C.c_int, c_double_p]),
    "bgp_get_factor_rows": (C.c_int, [handle_p, C.POINTER(C.c_int64), C.c_int, c_double_p]),
    "bgp_get_factor_diag": (C.c_int, [handle_p, c_double_p]),
    "bgp_phase_times": (C.c_int, [handle_p, c_double_p, C.c_int]),
    "bgp_device_bytes": (C.c_int64, [handle_p]),
    "bgp_potrf_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, c_int_p]),
    "bgp_gemm_nt_sub_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int],
    ),
    "bgp_fill_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double],
    ),
    "bgp_fill_block_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_double],
    ),
    "bgp_aug_rows_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]),
    "bgp_cross_block_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int64],
    ),
    "bgp_factor_panel_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, c_int_p]),
    "bgp_factor_pack_panel_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, c_int_p],
    ),
    "bgp_solve_panel_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p],
    ),
    "bgp_gemm_nt_sub_async_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int],
    ),
    "bgp_update_panels_dev": (C.c_int, [handle_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "bgp_factor_pack_panel_async_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    ),
    "bgp_flag_reset_dev": (C.c_int, [handle_p]),
    "bgp_flag_merge_dev": (C.c_int, [handle_p, C.c_void_p]),
    "bgp_flag_read": (C.c_int, [handle_p, c_int_p]),
    "bgp_get_stream": (C.c_void_p, [handle_p, C.c_int]),
    "bgp_gemm_nt_async_dev": (
        C.c_int,
        [handle_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int],
    ),
    "bgp_block_copy_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_int]),
    "bgp_panel_inverse_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "bgp_gemv_t_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "bgp_grad_nacc": (C.c_int, []),
    "bgp_grad_reduce_block_dev": (
        C.c_int,
        [handle_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int],
    ),
    "bgp_grad_finish": (C.c_int, [handle_p, c_double_p, C.c_int, c_double_p, C.c_int]),
    "bgp_diag_logsum_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, c_double_p]),
    "bgp_rowdot_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "bgp_var_finish_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double, C.c_void_p]),
    "bgp_debug_clock_samples_dev": (C.c_int, [handle_p, C.c_void_p, C.c_int, C.c_int]),
    "bgp_debug_set_ld_pad": (C.c_int, [handle_p, C.c_int64]),
    "bgp_sync": (C.c_int, [handle_p]),
}

# indices of bgp_phase_times (BGP_T_* in battgp.h)
(T_H2D, T_FILL, T_POTRF, T_SOLVE, T_CROSS, T_VAR, T_D2H, T_TRAIL, T_TRAIL_FLOP, T_FILL_BYTES, T_TRAIL_LAUNCHES, T_TRAIL_UNION,
 T_GRAD, T_RESTORE) = range(14)
T_COUNT = 14

_lib = None


def load() -> C.CDLL:
    """Load libbattgp.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m battgp_amd.build{' --experimental' if EXPERIMENTAL else ''}` "
            "(hipcc --offload-arch=gfx950).  battgp_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise ImportError(f"{os.path.basename(LIB_PATH)} does not export `{name}`") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(arr):
    """numpy fp64 C-contiguous array -> double*"""
    return arr.ctypes.data_as(c_double_p)
