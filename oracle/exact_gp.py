"""Exact-Cholesky GP algebra of the ``full_gp`` path (numpy + LAPACK, fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates what GPyTorch's Cholesky branch computes for the reference's call sites:

* ``psd_safe_cholesky`` - jitter ladder used by ``linear_operator`` (evidence of
  use in the reference: the ``NumericalWarning`` filter, ``gp_runner.py:14,28``).
* LML - ``ExactMarginalLogLikelihood`` as evaluated at
  ``src/gp/training.py:27-30,39-41``; the reference reports ``-mll * N``
  (``training.py:43``, ``src/batt_models/battcellgp_full.py:147,161-164``).
* posterior - ``ExactGP.__call__`` in eval mode as used at
  ``src/batt_models/battcellgp_full.py:171-180`` (latent ``f``: no noise is added
  to the predictive variance; ``.variance`` is floored at 1e-10) and
  ``src/gp/standard_models.py:40-48`` (unclamped covariance / its diagonal).
"""

from __future__ import annotations

import warnings

import numpy as np
import scipy.linalg as sla

from . import kernels as K


class NotPSDError(RuntimeError):
    """Mirrors ``linear_operator.utils.errors.NotPSDError``."""


class NumericalWarning(RuntimeWarning):
    """Mirrors ``linear_operator.utils.warnings.NumericalWarning``."""


JITTER_FP64 = 1e-8  # linear_operator.settings.cholesky_jitter, fp64 default
MAX_TRIES = 3  # linear_operator.settings.cholesky_max_tries default
MIN_VARIANCE_FP64 = 1e-10  # gpytorch.settings.min_variance, fp64 default


OPENBLAS_POTRF_LIMIT = 32768 - 1024  # this image's OpenBLAS (scipy's and numpy's) dpotrf segfaults from N = 2^15 on (seen at
                                     # 32 700; 30 000 is fine): from here on the factorisation goes through torch's LAPACK


def _dpotrf_lower(a: np.ndarray, overwrite: bool):
    """``(L, info)`` like ``scipy.linalg.lapack.dpotrf(a, lower=1, clean=1)``"""
    if a.shape[0] < OPENBLAS_POTRF_LIMIT:
        return sla.lapack.dpotrf(a, lower=1, clean=1, overwrite_a=int(overwrite))
    import torch

    fac, info = torch.linalg.cholesky_ex(torch.from_numpy(a))
    return fac.numpy(), int(info)


def psd_safe_cholesky(a: np.ndarray, jitter: float = JITTER_FP64, max_tries: int = MAX_TRIES):
    """Lower Cholesky factor with GPyTorch's jitter ladder.

    Plain ``potrf`` first; on failure add ``jitter * 10**i`` (``i = 0..max_tries-1``)
    to the diagonal of the *original* matrix, warn, and retry; after the last
    failure raise :class:`NotPSDError`.  Returns ``(L, jitter_used)``.
    """
    a = np.asarray(a, dtype=np.float64)
    if not np.all(np.isfinite(a)):
        raise NotPSDError("matrix contains NaN/inf")
    c, info = _dpotrf_lower(a, overwrite=False)
    if info == 0:
        return c, 0.0
    for i in range(max_tries):
        jit = jitter * (10.0**i)
        aj = a.copy()
        aj[np.diag_indices_from(aj)] += jit
        c, info = _dpotrf_lower(aj, overwrite=True)
        if info == 0:
            warnings.warn(
                f"A not p.d., added jitter of {jit:.1e} to the diagonal", NumericalWarning
            )
            return c, jit
    raise NotPSDError(
        f"Matrix not positive definite after repeatedly adding jitter up to {jit:.1e}."
    )


class OracleGP:
    """Zero-mean exact GP: ``fit`` once, then ``predict`` any number of times."""

    def __init__(self, kernel_id: int, hyp, x: np.ndarray, y: np.ndarray):
        self.kernel_id = int(kernel_id)
        self.hyp = np.asarray(hyp, dtype=np.float64).copy()
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.x = x.reshape(-1, 1) if x.ndim == 1 else x
        self.y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if self.x.shape[0] != self.y.shape[0]:
            raise ValueError("x and y disagree on N")
        self.L = None

    # -- fit -----------------------------------------------------------------
    def fit(self) -> "OracleGP":
        n = self.x.shape[0]
        # large N: same entries, evaluated block-wise on the host cores (bit-identical, see kernels.py)
        fill = K.kernel_matrix if n <= 2048 else K.kernel_matrix_blocked
        sigma = fill(self.kernel_id, self.hyp, self.x)
        sigma[np.diag_indices(n)] += K.noise(self.hyp)
        self.L, self.jitter = psd_safe_cholesky(sigma)
        # alpha = Sigma^-1 y  (mean cache of DefaultPredictionStrategy)
        self.z = sla.solve_triangular(self.L, self.y, lower=True)
        self.alpha = sla.solve_triangular(self.L, self.z, lower=True, trans="T")
        self.logdet_half = float(np.sum(np.log(np.diag(self.L))))
        self.lml = float(
            -0.5 * (self.z @ self.z) - self.logdet_half - 0.5 * n * np.log(2.0 * np.pi)
        )
        return self

    @property
    def neg_mll_scaled(self) -> float:
        """What the reference prints/saves as "Marginal Likelihood": ``-mll * N``
        with ``mll = lml / N`` (``training.py:43``)."""
        return -self.lml

    # -- predict ---------------------------------------------------------------
    def predict(self, xq: np.ndarray, full_cov: bool = False, clamp: bool = True):
        """Posterior of the latent ``f`` at ``xq``: ``(mean, var)`` or ``(mean, cov)``.

        ``clamp=True`` floors the variance at 1e-10 like ``MultivariateNormal.variance``
        (``battcellgp_full.py:180``); ``clamp=False`` is ``np.diag(out._covar)``
        (``standard_models.py:48``).
        """
        if self.L is None:
            self.fit()
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        if xq.ndim == 1:
            xq = xq.reshape(-1, 1)
        kxs = K.kernel_matrix(self.kernel_id, self.hyp, self.x, xq)  # [N, M]
        mean = kxs.T @ self.alpha
        v = sla.solve_triangular(self.L, kxs, lower=True)  # [N, M]
        if full_cov:
            cov = K.kernel_matrix(self.kernel_id, self.hyp, xq) - v.T @ v
            return mean, cov
        var = K.kernel_diag(self.kernel_id, self.hyp, xq) - np.einsum("ij,ij->j", v, v)
        if clamp:
            var = np.maximum(var, MIN_VARIANCE_FP64)
        return mean, var

    # -- on-device-style residual checks, restated ------------------------------
    def residuals(self) -> tuple[float, float]:
        """``||Sigma alpha - y|| / ||y||`` and ``||L L^T - Sigma||_F / ||Sigma||_F``."""
        n = self.x.shape[0]
        sigma = K.kernel_matrix(self.kernel_id, self.hyp, self.x)
        sigma[np.diag_indices(n)] += K.noise(self.hyp) + self.jitter
        r1 = np.linalg.norm(sigma @ self.alpha - self.y) / np.linalg.norm(self.y)
        r2 = np.linalg.norm(self.L @ self.L.T - sigma) / np.linalg.norm(sigma)
        return float(r1), float(r2)


def kernel_derivatives(kernel_id: int, hyp, x1: np.ndarray, x2: np.ndarray) -> list[np.ndarray]:
    """``[dK(x1, x2)/d hyp_i for i = 1 .. len(hyp) - 1]`` (the noise term, i = 0, is the identity on the training
    diagonal and is left to the caller) - the kernel derivatives autograd differentiates through at
    ``src/gp/training.py:41``; block form so that a distributed reduction can be checked panel by panel."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x1 = np.atleast_2d(np.asarray(x1, dtype=np.float64))
    x2 = np.atleast_2d(np.asarray(x2, dtype=np.float64))
    d = x1.shape[1]

    def sq(col, ls_d):
        diff = (x1[:, col : col + 1] - x2[:, col : col + 1].T) / ls_d
        return diff * diff

    if kernel_id == K.KERNEL_BATTGP:
        s_r, ls = hyp[2], hyp[3:]
        e = np.exp(-0.5 * K._scaled_sqdist(x1[:, 1:], x2[:, 1:], ls))
        return [K.integrated_wiener(x1[:, 0], x2[:, 0]), e] + [s_r * e * sq(1 + dd, ls[dd]) / ls[dd] for dd in range(d - 1)]
    if kernel_id == K.KERNEL_SCALED_RBF:
        s, ell = hyp[1], hyp[2]
        q = K._scaled_sqdist(x1, x2, np.full(d, ell))
        e = np.exp(-0.5 * q)
        return [e, s * e * q / ell]
    if kernel_id == K.KERNEL_ARD_RBF:
        s, ls = hyp[1], hyp[2:]
        e = np.exp(-0.5 * K._scaled_sqdist(x1, x2, ls))
        return [e] + [s * e * sq(dd, ls[dd]) / ls[dd] for dd in range(d)]
    if kernel_id == K.KERNEL_MATERN32:
        s, ls = hyp[1], hyp[2:]
        a = np.sqrt(3.0) * np.sqrt(K._scaled_sqdist(x1, x2, ls))
        ea = np.exp(-a)
        # dk/dl_d = s * 3 * exp(-a) * ((x_d-x'_d)/l_d)^2 / l_d
        return [(1.0 + a) * ea] + [s * 3.0 * ea * sq(dd, ls[dd]) / ls[dd] for dd in range(d)]
    raise ValueError(f"unknown kernel id {kernel_id}")


def lml_and_grad(kernel_id: int, hyp, x: np.ndarray, y: np.ndarray):
    """LML and its gradient w.r.t. the hyper-parameter vector (same layout as
    ``hyp``):  ``d lml / d theta = 1/2 tr((alpha alpha^T - Sigma^-1) dSigma/dtheta)``.

    This is what autograd produces for ``loss.backward()`` at
    ``src/gp/training.py:41`` up to the factor ``-1/N`` and the raw-parameter
    transforms; used to check the engine's gradient kernels on small N.
    """
    hyp = np.asarray(hyp, dtype=np.float64)
    gp = OracleGP(kernel_id, hyp, x, y).fit()
    xx = gp.x
    n = xx.shape[0]
    linv = sla.solve_triangular(gp.L, np.eye(n), lower=True)
    w = np.outer(gp.alpha, gp.alpha) - linv.T @ linv  # alpha alpha^T - Sigma^-1
    grad = np.zeros_like(hyp)
    grad[0] = 0.5 * np.trace(w)  # dSigma/dnoise = I
    for i, dk in enumerate(kernel_derivatives(kernel_id, hyp, xx, xx)):
        grad[1 + i] = 0.5 * np.sum(w * dk)
    return gp.lml, grad
