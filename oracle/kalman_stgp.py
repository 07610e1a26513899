"""Kalman-filter spatio-temporal GP, restated, as an INDEPENDENT cross-check of
the exact GP with the production kernel family (integrated Wiener + ARD-RBF).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``src/gp/spatiotemporal_gp.py``: state prediction ``:20-29``, state
layout/initialisation ``:104-118``, output matrix ``:150-173``, predictive
covariance ``:175-179``, measurement update ``:181-202``, ``predict`` ``:204-216``.
The temporal model ``(A, Q)`` is injected (``kalman_matrices``) so that the golden
generator can drive it with the reference's own, importable
``WienerTemporalKernel._get_kalman_matrices``
(``src/gp/wiener_kernel_temporal.py:28-35``); :func:`wiener_kalman_matrices` is
the restated fallback checked against it.

It reproduces the exact GP only when every training input's spatial part is one
of the basis vectors (``tests/gp/test_spatiotemporal_gp.py:218-222``).
"""

from __future__ import annotations

from typing import Callable

import numpy as np

from . import kernels as K


def wiener_kalman_matrices(outputscale: float, ts: float):
    """``A = [[1,Ts],[0,1]]``, ``Q = s [[Ts^3/3, Ts^2/2],[Ts^2/2, Ts]]``
    (``src/gp/wiener_kernel_temporal.py:32-33``); ``Ts == 0`` gives ``(I, 0)``
    (``src/gp/temporal_kernel.py:23-25``)."""
    if ts == 0.0:
        return np.eye(2), np.zeros((2, 2))
    a = np.array([[1.0, ts], [0.0, 1.0]])
    q = outputscale * np.array([[ts**3 / 3.0, ts**2 / 2.0], [ts**2 / 2.0, ts]])
    return a, q


class KalmanSTGP:
    def __init__(
        self,
        s_base: np.ndarray,
        rbf_hyp: np.ndarray,  # KERNEL_ARD_RBF layout [unused noise, s, l_1..l_D]
        noise_var: float,
        kalman_matrices: Callable[[float], tuple[np.ndarray, np.ndarray]],
    ):
        self.s_base = np.ascontiguousarray(s_base, dtype=np.float64)
        self.rbf_hyp = np.asarray(rbf_hyp, dtype=np.float64)
        self.noise_var = float(noise_var)
        self.kalman_matrices = kalman_matrices
        self.kbb = K.kernel_matrix(K.KERNEL_ARD_RBF, self.rbf_hyp, self.s_base)
        self.inv_kbb = np.linalg.pinv(self.kbb, rcond=1e-8)
        nb = self.s_base.shape[0]
        self.nt = 2
        self.P = np.zeros((2 + nb, 2 + nb))
        self.P[2:, 2:] = self.kbb
        self.z = np.zeros((2 + nb, 1))
        self.t = 0.0

    def time_step(self, dt: float) -> None:
        a, q = self.kalman_matrices(dt)
        nt = self.nt
        self.z[:nt, :] = a @ self.z[:nt, :]
        self.P[:nt, :nt] = a @ self.P[:nt, :nt] @ a.T + q
        self.P[nt:, :nt] = self.P[nt:, :nt] @ a.T
        self.P[:nt, nt:] = a @ self.P[:nt, nt:]
        self.t += dt

    def _output_matrix(self, xs: np.ndarray):
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ktb = K.kernel_matrix(K.KERNEL_ARD_RBF, self.rbf_hyp, xs, self.s_base)
        ktt = K.kernel_matrix(K.KERNEL_ARD_RBF, self.rbf_hyp, xs)
        h = np.zeros((xs.shape[0], self.nt + self.s_base.shape[0]))
        h[:, 0] = 1.0
        h[:, self.nt :] = ktb @ self.inv_kbb
        return h, ktt

    def _covariance(self, h, ktt, p):
        hs = h[:, self.nt :]
        return ktt + h @ p @ h.T - hs @ self.kbb @ hs.T

    def update(self, xs: np.ndarray, y: np.ndarray) -> float:
        """Measurement update; returns the innovation log-likelihood
        ``log N(y | H z, C + sigma^2 I)`` of this batch.  Summed over the filter run it is the joint
        ``log p(y_1..n)`` - the exact GP's log-marginal likelihood when the equivalence conditions of
        ``tests/gp/test_spatiotemporal_gp.py:218-222`` hold - which pins the LML independently of any
        Cholesky of the full covariance."""
        h, ktt = self._output_matrix(xs)
        c = self._covariance(h, ktt, self.P)
        v = np.asarray(y, dtype=np.float64).reshape(-1, 1) - h @ self.z
        s = c + np.eye(xs.shape[0]) * self.noise_var
        s_inv = np.linalg.inv(s)
        gain = self.P @ h.T @ s_inv
        self.z = self.z + gain @ v
        self.P = self.P - gain @ h @ self.P
        _, logdet = np.linalg.slogdet(s)
        return float(-0.5 * (v.T @ s_inv @ v).item() - 0.5 * logdet - 0.5 * xs.shape[0] * np.log(2.0 * np.pi))

    def predict(self, xq: np.ndarray):
        h, kqq = self._output_matrix(xq)
        c = self._covariance(h, kqq, self.P)
        return (h @ self.z).reshape(-1), np.diag(c).copy()
