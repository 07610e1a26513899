"""Covariance functions of the ``full_gp`` path, restated in numpy (fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Hyper-parameter vector layout (identical to the C-ABI, ``include/battgp.h``):

* ``KERNEL_BATTGP``     ``[noise, s_wiener, s_rbf, l_1 .. l_{D-1}]`` - column 0
  of ``X`` is time (integrated Wiener), columns 1..D-1 are the ARD-RBF inputs.
* ``KERNEL_SCALED_RBF`` ``[noise, s, l]`` - isotropic over all D columns.
* ``KERNEL_MATERN32``   ``[noise, s, l_1 .. l_D]`` - ARD Matern-3/2 over all D
  columns (no reference call site; GPyTorch ``MaternKernel(nu=1.5)`` formula).
* ``KERNEL_ARD_RBF``    ``[noise, s, l_1 .. l_D]`` - ARD RBF over all D columns
  (the spatial kernel of ``tests/gp/test_spatiotemporal_gp.py:65-81``).
"""

from __future__ import annotations

import numpy as np

KERNEL_BATTGP = 0
KERNEL_SCALED_RBF = 1
KERNEL_MATERN32 = 2
KERNEL_ARD_RBF = 3


def n_hyp(kernel_id: int, d: int) -> int:
    if kernel_id == KERNEL_BATTGP:
        return 3 + (d - 1)
    if kernel_id == KERNEL_SCALED_RBF:
        return 3
    if kernel_id in (KERNEL_MATERN32, KERNEL_ARD_RBF):
        return 2 + d
    raise ValueError(f"unknown kernel id {kernel_id}")


def integrated_wiener(t1: np.ndarray, t2: np.ndarray) -> np.ndarray:
    """``k_IW(s,t) = min(s,t)^3/3 + |s-t| min(s,t)^2/2``.

    Follows ``src/gp/wiener_kernel.py:10-32`` (the non-``diag`` branch: ``minval``
    is the pairwise minimum, ``distance`` the pairwise absolute difference).
    GPyTorch's ``covar_dist`` floors the distance at ``sqrt(1e-30)`` on coincident
    points; that 1e-15 term is below fp64 resolution of the result and is dropped.
    """
    a = np.asarray(t1, dtype=np.float64).reshape(-1, 1)
    b = np.asarray(t2, dtype=np.float64).reshape(1, -1)
    m = np.minimum(a, b)
    return m * m * m / 3.0 + np.abs(a - b) * (m * m) / 2.0


def _scaled_sqdist(x1: np.ndarray, x2: np.ndarray, ls: np.ndarray) -> np.ndarray:
    """``sum_d ((x1_d - x2_d)/l_d)^2`` by direct differences (no quadratic
    expansion), accumulated dimension by dimension to keep memory at O(n m)."""
    out = np.zeros((x1.shape[0], x2.shape[0]), dtype=np.float64)
    for d in range(x1.shape[1]):
        diff = (x1[:, d : d + 1] - x2[:, d : d + 1].T) / ls[d]
        out += diff * diff
    return out


def kernel_matrix(
    kernel_id: int, hyp: np.ndarray, x1: np.ndarray, x2: np.ndarray | None = None
) -> np.ndarray:
    """Noise-free prior covariance ``K(x1, x2)``; ``x2=None`` means ``x1``.

    K0: ``ScaleKernel(WienerKernel(active_dims=[0])) +
    ScaleKernel(RBFKernel(ard_num_dims=D-1, active_dims=[1..D-1]))``
    (``src/batt_models/cell_gp.py:32-36``).
    K1: ``ScaleKernel(RBFKernel())`` (``src/gp/standard_models.py:24``).
    """
    hyp = np.asarray(hyp, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    if x1.ndim == 1:
        x1 = x1.reshape(-1, 1)
    x2 = x1 if x2 is None else np.ascontiguousarray(x2, dtype=np.float64)
    if x2.ndim == 1:
        x2 = x2.reshape(-1, 1)
    d = x1.shape[1]
    if hyp.shape[0] != n_hyp(kernel_id, d):
        raise ValueError("hyper-parameter vector has the wrong length")
    if kernel_id == KERNEL_BATTGP:
        s_w, s_r, ls = hyp[1], hyp[2], hyp[3:]
        k = s_w * integrated_wiener(x1[:, 0], x2[:, 0])
        k += s_r * np.exp(-0.5 * _scaled_sqdist(x1[:, 1:], x2[:, 1:], ls))
        return k
    if kernel_id == KERNEL_SCALED_RBF:
        s, ell = hyp[1], hyp[2]
        return s * np.exp(-0.5 * _scaled_sqdist(x1, x2, np.full(d, ell)))
    if kernel_id == KERNEL_ARD_RBF:
        s, ls = hyp[1], hyp[2:]
        return s * np.exp(-0.5 * _scaled_sqdist(x1, x2, ls))
    if kernel_id == KERNEL_MATERN32:
        s, ls = hyp[1], hyp[2:]
        r = np.sqrt(_scaled_sqdist(x1, x2, ls))
        a = np.sqrt(3.0) * r
        return s * (1.0 + a) * np.exp(-a)
    raise ValueError(f"unknown kernel id {kernel_id}")


def kernel_matrix_blocked(
    kernel_id: int, hyp: np.ndarray, x: np.ndarray, block: int = 1024, threads: int | None = None
) -> np.ndarray:
    """``kernel_matrix(kernel_id, hyp, x)`` for large N: the lower-triangle blocks are evaluated by
    :func:`kernel_matrix` on a thread pool (numpy ufuncs release the GIL) and mirrored.  Every entry goes through
    exactly the arithmetic of :func:`kernel_matrix` (all operations are elementwise), so the result is
    bit-identical to the one-shot evaluation - it only keeps the temporaries small and uses the host cores."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n = x.shape[0]
    out = np.empty((n, n), dtype=np.float64)
    starts = list(range(0, n, block))
    jobs = [(i0, j0) for i0 in starts for j0 in starts if j0 <= i0]

    def run(job):
        i0, j0 = job
        i1, j1 = min(i0 + block, n), min(j0 + block, n)
        blk = kernel_matrix(kernel_id, hyp, x[i0:i1], x[j0:j1])
        out[i0:i1, j0:j1] = blk
        if i0 != j0:
            out[j0:j1, i0:i1] = blk.T

    workers = threads or min(32, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(run, jobs))
    return out


def kernel_diag(kernel_id: int, hyp: np.ndarray, x: np.ndarray) -> np.ndarray:
    """``diag K(x, x)`` (noise-free).  K0 follows the ``diag=True`` branch of
    ``src/gp/wiener_kernel.py:15-16`` (``min = t`` => ``t^3/3``)."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    if kernel_id == KERNEL_BATTGP:
        t = x[:, 0]
        return hyp[1] * (t * t * t / 3.0) + hyp[2]
    return np.full(x.shape[0], hyp[1], dtype=np.float64)


def noise(hyp: np.ndarray) -> float:
    return float(np.asarray(hyp, dtype=np.float64)[0])
