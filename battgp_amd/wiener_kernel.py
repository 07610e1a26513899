"""Dense evaluation of the path's covariance functions on the GPU, mirroring what
``kernel(x1, x2).to_dense()`` returns in the reference (``WienerKernel.forward``,
``src/gp/wiener_kernel.py:10-32``; composition ``src/batt_models/cell_gp.py:32-36``)."""

from __future__ import annotations

import numpy as np

from . import KERNEL_BATTGP
from .engine import ExactGPEngine


class WienerRBFKernel:
    """``ScaleKernel(WienerKernel(active_dims=[0])) + ScaleKernel(RBFKernel(ard, active_dims=[1..]))``
    evaluated by the same fused HIP fill kernel the fit uses."""

    def __init__(self, outputscale_wiener, outputscale_rbf, lengthscale_rbf, device=0):
        ls = np.atleast_1d(np.asarray(lengthscale_rbf, dtype=np.float64))
        self._ls = ls
        self._sw, self._sr = float(outputscale_wiener), float(outputscale_rbf)
        self._device = device
        self._engine = None

    def __call__(self, x1: np.ndarray, x2: np.ndarray | None = None) -> np.ndarray:
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        d1 = x1.shape[1] - 1
        ls = np.full(d1, self._ls[0]) if self._ls.size == 1 else self._ls
        hyp = np.concatenate(([0.0, self._sw, self._sr], ls))
        if self._engine is None:
            self._engine = ExactGPEngine(KERNEL_BATTGP, hyp, device=self._device)
        else:
            self._engine.set_hyp(hyp)
        return self._engine.kernel_matrix(x1, x2)

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
