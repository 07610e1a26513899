"""``ScaledRBFModel`` - isotropic ``s * exp(-|x-x'|^2 / (2 l^2))`` exact GP with the surface of
``src/gp/standard_models.py:8-55`` (the only exact-GP class the reference's unit tests pin:
``tests/gp/test_standard_models.py``, ``tests/gp/test_recursive_gp.py:195-232``).

The reference casts data to float32 (``:19-20,41``); this engine computes in fp64 and returns fp64
- its answers agree with the reference's to the reference's own fp32 test tolerances.
``predict`` returns the UNCLAMPED covariance diagonal (``np.diag(out._covar)``, ``:48``) or the
full ``[M, M]`` posterior covariance (``:45-46``).
"""

from __future__ import annotations

import numpy as np
import torch

from . import KERNEL_SCALED_RBF
from .engine import ExactGPEngine, as_device_index


class ScaledRBFModel:
    def __init__(self, train_x, train_y, noise_variance, outputscale, lengthscale, device=None):
        x = torch.as_tensor(train_x, dtype=torch.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = torch.as_tensor(train_y, dtype=torch.float64).reshape(-1)
        self.train_inputs = (x,)
        self.train_targets = y
        self._hyp = np.array([float(noise_variance), float(outputscale), float(lengthscale)], dtype=np.float64)
        self._device = device
        self._engine: ExactGPEngine | None = None
        self._fitted = False
        self.lml = None

    def _fit(self):
        if self._engine is None:
            self._engine = ExactGPEngine(KERNEL_SCALED_RBF, self._hyp, device=as_device_index(self._device))
        if not self._fitted:
            self._engine.set_hyp(self._hyp)
            self.lml = self._engine.fit(self.train_inputs[0].cpu().numpy(), self.train_targets.cpu().numpy())
            self._fitted = True
        return self._engine

    def predict(self, xq: np.ndarray, full_cov: bool = False) -> tuple[np.ndarray, np.ndarray]:
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        if xq.ndim == 1:
            xq = xq.reshape(-1, self.train_inputs[0].shape[1])
        eng = self._fit()
        if full_cov:
            return eng.predict_cov(xq)
        return eng.predict(xq, want_var=True, min_var=-1.0)

    def optimize(self, **kwargs):
        # the reference calls training.train_exact_gp, which does not exist in src/gp/training.py
        # (AttributeError upstream); keep the failure explicit
        raise AttributeError("module 'training' has no attribute 'train_exact_gp' (dangling in the reference too)")

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
            self._fitted = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
