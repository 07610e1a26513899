"""ExactGPEngine - thin Python owner of one ``bgp_handle`` (one GP on one MI355X).

numpy in / numpy out, like the reference's ``IBatteryCellGP`` surface
(``src/batt_models/batt_cell_gp_protocol.py:9-86``).  All arithmetic happens in
libbattgp.so; this file only marshals pointers and maps return codes to the
exceptions/warnings GPyTorch raises for the same conditions.
"""

from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import dptr


class EngineError(RuntimeError):
    """HIP / allocation / argument error reported by libbattgp.so."""


class NotPSDError(RuntimeError):
    """Counterpart of ``linear_operator.utils.errors.NotPSDError`` (jitter ladder exhausted)."""


class NumericalWarning(RuntimeWarning):
    """Counterpart of ``linear_operator.utils.warnings.NumericalWarning`` (jitter was added);
    the reference silences it at ``gp_runner.py:28``."""


def as_device_index(device) -> int:
    """Accept what the reference passes as ``device=``: int (``gp_runner.py:162-171``),
    ``torch.device``, or a string like ``"cuda:1"``.  CPU devices are rejected: this engine
    has no CPU path."""
    if device is None:
        return 0
    if isinstance(device, (int, np.integer)):
        return int(device)
    s = str(device)
    if s.startswith("cpu"):
        raise EngineError(
            "battgp_amd runs on MI355X only; pass device=<gpu index> or torch.device('cuda:N')"
        )
    if ":" in s:
        return int(s.split(":")[1])
    return 0


def trim_pool(device: int = -1) -> None:
    """Release the HBM of parked (destroyed) engine handles - ``torch.cuda.empty_cache()`` for this engine."""
    _lib.load().bgp_trim(int(device))


class ExactGPEngine:
    def __init__(self, kernel_id: int, hyp, device=0):
        self._lib = _lib.load()
        self._h = _lib.handle_p()
        self.device_index = as_device_index(device)
        rc = self._lib.bgp_create(C.byref(self._h), self.device_index)
        if rc != 0:
            msg = self._lib.bgp_last_error(None).decode()
            self._h = None
            raise EngineError(f"bgp_create failed ({rc}): {msg}")
        self.kernel_id = int(kernel_id)
        self.n = 0
        self.d = 0
        self.lml = None
        self.jitter = None
        self.set_hyp(hyp)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.bgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc == 0:
            return
        msg = self._lib.bgp_last_error(self._h).decode()
        if rc > 0:
            raise NotPSDError(f"{what}: {msg}")
        raise EngineError(f"{what} failed ({rc}): {msg}")

    # -- configuration ------------------------------------------------------------------------
    def set_hyp(self, hyp) -> None:
        hyp = np.ascontiguousarray(np.asarray(hyp, dtype=np.float64).reshape(-1))
        self.hyp = hyp.copy()
        self._check(self._lib.bgp_set_kernel(self._h, self.kernel_id, dptr(hyp), hyp.size), "bgp_set_kernel")

    def set_options(self, nb_outer=-1, max_tries=-1, jitter0=-1.0, lookahead=-1) -> None:
        self._check(self._lib.bgp_set_options(self._h, nb_outer, max_tries, jitter0, lookahead), "bgp_set_options")

    def set_panel_scheme(self, scheme: int = -1) -> None:
        """1: critical chain on the diagonal block + one deep TRSM-by-inverse GEMM; 0: 64-wide chain over all rows;
        -1 (default): by size."""
        self._check(self._lib.bgp_set_panel_scheme(self._h, int(scheme)), "bgp_set_panel_scheme")

    def set_layout(self, slab_width: int = 0) -> None:
        """HBM layout of the factor: -1 full square (8 N^2 B), > 0 column slabs of that width
        (~4 N (N + W) B, what lets N = 262 144 fit one MI355X), 0 = automatic (default)."""
        self._check(self._lib.bgp_set_layout(self._h, int(slab_width)), "bgp_set_layout")

    def set_keep_factor(self, on: bool = True) -> None:
        """Opt-in: ``lml_grad`` saves the factor into a second buffer first (when memory allows) and the next call that
        needs it copies it back instead of re-running the fit (``bgp_set_keep_factor``)."""
        self._check(self._lib.bgp_set_keep_factor(self._h, int(bool(on))), "bgp_set_keep_factor")

    def debug_set_ld_pad(self, extra_rows: int) -> None:
        """Tests / diagnostics: unused rows appended to every column of the factor buffer from the next fit on
        (``bgp_debug_set_ld_pad``): BASELINE-size element strides on a small problem."""
        self._check(self._lib.bgp_debug_set_ld_pad(self._h, int(extra_rows)), "bgp_debug_set_ld_pad")

    def layout(self) -> tuple[int, int]:
        """(slab width in use, 0 = full square; bytes of the factor buffer)."""
        w, b = C.c_int64(0), C.c_int64(0)
        self._lib.bgp_get_layout(self._h, C.byref(w), C.byref(b))
        return int(w.value), int(b.value)

    # -- fit / predict ------------------------------------------------------------------------
    def _after_fit(self, lml, jit):
        self.lml = float(lml.value)
        self.jitter = float(jit.value)
        if self.jitter > 0.0:
            warnings.warn(f"A not p.d., added jitter of {self.jitter:.1e} to the diagonal", NumericalWarning)
        return self.lml

    def fit(self, x: np.ndarray, y: np.ndarray) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if x.shape[0] != y.shape[0]:
            raise ValueError("x and y disagree on N")
        n, d = x.shape
        lml, jit = C.c_double(), C.c_double()
        rc = self._lib.bgp_fit(self._h, dptr(x), dptr(y), n, d, C.byref(lml), C.byref(jit))
        self._check(rc, "bgp_fit")
        self.n, self.d = n, d  # only a successful call changes what the engine holds
        return self._after_fit(lml, jit)

    def fit_device(self, x_ptr: int, y_ptr: int, n: int, d: int) -> float:
        """X[n,d], y[n] already resident on this engine's GPU (e.g. ``tensor.data_ptr()``)."""
        lml, jit = C.c_double(), C.c_double()
        rc = self._lib.bgp_fit_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), int(n), int(d), C.byref(lml), C.byref(jit))
        self._check(rc, "bgp_fit_dev")
        self.n, self.d = int(n), int(d)
        return self._after_fit(lml, jit)

    def fit_predict(self, x: np.ndarray, y: np.ndarray, xq: np.ndarray, want_var: bool = True, min_var: float = 1e-10):
        """Fit and evaluate the posterior at ``xq`` in one pass (the query rows ride through the
        factorisation).  Returns ``(lml, mean, var)`` (``var`` is None if not wanted)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        xq = np.ascontiguousarray(xq, dtype=np.float64).reshape(-1, x.shape[1])
        n, d = x.shape
        m = xq.shape[0]
        mean = np.empty(m, dtype=np.float64)
        var = np.empty(m, dtype=np.float64) if want_var else None
        lml, jit = C.c_double(), C.c_double()
        rc = self._lib.bgp_fit_predict(
            self._h, dptr(x), dptr(y), n, d, dptr(xq), m, C.byref(lml), C.byref(jit), dptr(mean),
            dptr(var) if want_var else None, float(min_var),
        )
        self._check(rc, "bgp_fit_predict")
        self.n, self.d = n, d
        return self._after_fit(lml, jit), mean, var

    def fit_predict_device(self, x_ptr, y_ptr, n, d, xq_ptr, m, mean_ptr, var_ptr, min_var: float = 1e-10) -> float:
        lml, jit = C.c_double(), C.c_double()
        rc = self._lib.bgp_fit_predict_dev(
            self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), int(n), int(d), C.c_void_p(xq_ptr), int(m), C.byref(lml),
            C.byref(jit), C.c_void_p(mean_ptr), C.c_void_p(var_ptr) if var_ptr else None, float(min_var),
        )
        self._check(rc, "bgp_fit_predict_dev")
        self.n, self.d = int(n), int(d)
        return self._after_fit(lml, jit)

    def refit(self, hyp) -> float:
        hyp = np.ascontiguousarray(np.asarray(hyp, dtype=np.float64).reshape(-1))
        lml, jit = C.c_double(), C.c_double()
        rc = self._lib.bgp_refit(self._h, dptr(hyp), hyp.size, C.byref(lml), C.byref(jit))
        self._check(rc, "bgp_refit")
        self.hyp = hyp.copy()
        return self._after_fit(lml, jit)

    def lml_grad(self) -> np.ndarray:
        """d lml / d hyp (same layout as ``hyp``) at the last fit, computed on the GPU.  ``Sigma^-1`` is formed in place
        over the factor (no second N^2 buffer); a later ``predict`` / ``residuals`` / ``lml_grad`` at the same point
        transparently re-runs the fit on the resident data (``include/battgp.h``)."""
        g = np.zeros(self.hyp.size, dtype=np.float64)
        self._check(self._lib.bgp_lml_grad(self._h, dptr(g), g.size), "bgp_lml_grad")
        return g

    def predict(self, xq: np.ndarray, want_var: bool = True, min_var: float = 1e-10):
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        if self.d == 0:
            raise EngineError("bgp_predict: no successful fit on this handle")
        if xq.ndim == 1:
            xq = xq.reshape(-1, self.d)
        if xq.shape[1] != self.d:
            raise ValueError(f"query has {xq.shape[1]} columns, model has {self.d}")
        m = xq.shape[0]
        mean = np.empty(m, dtype=np.float64)
        var = np.empty(m, dtype=np.float64) if want_var else None
        rc = self._lib.bgp_predict(
            self._h, dptr(xq), m, dptr(mean), dptr(var) if want_var else None, float(min_var)
        )
        self._check(rc, "bgp_predict")
        return (mean, var) if want_var else mean

    def predict_device(self, xq_ptr: int, m: int, mean_ptr: int, var_ptr, min_var: float = 1e-10):
        rc = self._lib.bgp_predict_dev(
            self._h,
            C.c_void_p(xq_ptr),
            int(m),
            C.c_void_p(mean_ptr),
            C.c_void_p(var_ptr) if var_ptr else None,
            float(min_var),
        )
        self._check(rc, "bgp_predict_dev")

    def predict_cov(self, xq: np.ndarray):
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        if xq.ndim == 1:
            xq = xq.reshape(-1, self.d)
        m = xq.shape[0]
        mean = np.empty(m, dtype=np.float64)
        cov = np.empty((m, m), dtype=np.float64)
        self._check(self._lib.bgp_predict_cov(self._h, dptr(xq), m, dptr(mean), dptr(cov)), "bgp_predict_cov")
        return mean, cov

    def kernel_matrix(self, x1: np.ndarray, x2: np.ndarray | None = None) -> np.ndarray:
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        if x1.ndim == 1:
            x1 = x1.reshape(-1, 1)
        if x2 is None:
            n2, p2 = x1.shape[0], None
        else:
            x2 = np.ascontiguousarray(x2, dtype=np.float64)
            if x2.ndim == 1:
                x2 = x2.reshape(-1, 1)
            n2, p2 = x2.shape[0], dptr(x2)
        out = np.empty((x1.shape[0], n2), dtype=np.float64)
        rc = self._lib.bgp_kernel_matrix(self._h, dptr(x1), x1.shape[0], p2, n2, x1.shape[1], dptr(out))
        self._check(rc, "bgp_kernel_matrix")
        return out

    def alpha(self) -> np.ndarray:
        a = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.bgp_get_alpha(self._h, dptr(a)), "bgp_get_alpha")
        return a

    def factor_rows(self, rows) -> np.ndarray:
        """Rows of the Cholesky factor of the last fit, ``out[r, q] = L[rows[r], q]`` (zero for q > rows[r])."""
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1))
        out = np.empty((rows.size, self.n), dtype=np.float64)
        rc = self._lib.bgp_get_factor_rows(self._h, rows.ctypes.data_as(C.POINTER(C.c_int64)), rows.size, dptr(out))
        self._check(rc, "bgp_get_factor_rows")
        return out

    def factor_diag(self) -> np.ndarray:
        d = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.bgp_get_factor_diag(self._h, dptr(d)), "bgp_get_factor_diag")
        return d

    def residuals(self, nsample: int = 256):
        out = np.zeros(2, dtype=np.float64)
        self._check(self._lib.bgp_residuals(self._h, int(nsample), dptr(out)), "bgp_residuals")
        return float(out[0]), float(out[1])

    def phase_times(self) -> dict:
        t = np.zeros(_lib.T_COUNT, dtype=np.float64)
        self._lib.bgp_phase_times(self._h, dptr(t), _lib.T_COUNT)
        names = [
            "h2d_ms", "fill_ms", "potrf_ms", "solve_ms", "cross_ms", "var_ms", "d2h_ms",
            "trail_ms", "trail_flop", "fill_bytes", "trail_launches", "trail_union_ms", "grad_ms", "restore_ms",
        ]
        return dict(zip(names, (float(v) for v in t)))

    def device_bytes(self) -> int:
        return int(self._lib.bgp_device_bytes(self._h))

    # -- building blocks on raw device pointers (tests, bench roofline, sharded driver) ----------
    def potrf_device(self, a_ptr: int, n: int, lda: int) -> int:
        info = C.c_int(0)
        self._check(self._lib.bgp_potrf_dev(self._h, C.c_void_p(a_ptr), n, lda, C.byref(info)), "bgp_potrf_dev")
        return int(info.value)

    def gemm_nt_sub_device(self, c_ptr, ldc, a_ptr, lda, b_ptr, ldb, m, n, k, lower=0) -> None:
        rc = self._lib.bgp_gemm_nt_sub_dev(
            self._h, C.c_void_p(c_ptr), ldc, C.c_void_p(a_ptr), lda, C.c_void_p(b_ptr), ldb, m, n, k, int(lower)
        )
        self._check(rc, "bgp_gemm_nt_sub_dev")

    def fill_block_device(self, x_ptr, n, d, row0, col0, nrows, ncols, out_ptr, ld, extra_diag=0.0) -> None:
        """Block [row0, row0+nrows) x [col0, col0+ncols) of Sigma = K(X, X) + (noise + extra_diag) I into a
        column-major device buffer (asynchronous on the engine's stream: call :meth:`sync`)."""
        rc = self._lib.bgp_fill_block_dev(
            self._h, C.c_void_p(x_ptr), int(n), int(d), int(row0), int(col0), int(nrows), int(ncols), C.c_void_p(out_ptr), int(ld),
            float(extra_diag),
        )
        self._check(rc, "bgp_fill_block_dev")

    def sync(self) -> None:
        self._check(self._lib.bgp_sync(self._h), "bgp_sync")

    def fill_device(self, x1_ptr, n1, x2_ptr, n2, d, out_ptr, ld, lower=0, diag_add=0.0) -> None:
        rc = self._lib.bgp_fill_dev(
            self._h, C.c_void_p(x1_ptr), n1, C.c_void_p(x2_ptr), n2, d, C.c_void_p(out_ptr), ld, int(lower), float(diag_add)
        )
        self._check(rc, "bgp_fill_dev")
