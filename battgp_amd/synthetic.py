"""Deterministic synthetic battery inputs shaped like ``BattData.generateTrainingData``
output (``src/batt_data/batt_data.py:180-256`` of the reference): ``X[N,4]`` fp64 with
columns ``(time[days], I[A], SOC[%], T[degC])`` and ``y[N]`` resistance in Ohm.

Ranges follow the reference's segment filters (``src/config.py:99-111``) and time is
reset to start at 0 (``batt_data.py:90-94``).  Used by tests, ``bench.py`` and the
smoke entry; there is no network, so the field data set itself is not available.
"""

from __future__ import annotations

import numpy as np

# production hyper-parameters, src/config.py:39-43 (exact fp64 literals)
NOISE_VARIANCE = 2.33e-6
OUTPUTSCALE_WIENER = 4.23e-13
OUTPUTSCALE_RBF = 0.0099
LENGTHSCALE_RBF = (12.11, 33.75, 45.14)

HYP_BATTGP = np.array(
    [NOISE_VARIANCE, OUTPUTSCALE_WIENER, OUTPUTSCALE_RBF, *LENGTHSCALE_RBF], dtype=np.float64
)
# Matern-3/2 ARD over (t, I, SOC, T): BASELINE config 3 (no reference call site)
HYP_MATERN32 = np.array(
    [NOISE_VARIANCE, OUTPUTSCALE_RBF, 400.0, *LENGTHSCALE_RBF], dtype=np.float64
)
# reference operating point, gp_runner.py:32  (I, SOC, T)
REF_OP = (-15.0, 90.0, 25.0)
N_QUERY = 300  # battgp_full.py:98


def make_cell_data(n: int, seed: int | None = None, age_days: float = 1200.0):
    """``(X[N,4], y[N])`` for one cell; ``seed`` defaults to ``n``."""
    rng = np.random.default_rng(n if seed is None else seed)
    t = np.sort(rng.uniform(0.0, age_days, n))
    t[0] = 0.0
    cur = rng.uniform(-80.0, -5.0, n)
    soc = rng.uniform(40.0, 95.0, n)
    temp = rng.uniform(10.0, 45.0, n)
    y = (
        0.012
        + 0.002 * np.exp(-(temp - 10.0) / 20.0)
        + 1e-6 * t
        + rng.normal(0.0, np.sqrt(NOISE_VARIANCE), n)
    )
    x = np.ascontiguousarray(np.stack([t, cur, soc, temp], axis=1), dtype=np.float64)
    return x, np.ascontiguousarray(y, dtype=np.float64)


def make_query(x: np.ndarray, m: int = N_QUERY, op=REF_OP) -> np.ndarray:
    """300-point time grid at the reference operating point
    (``battgp_full.py:98`` + ``battcellgp_full.py:199-206``)."""
    t = np.linspace(x[0, 0], x[-1, 0], m)
    return np.ascontiguousarray(
        np.column_stack((t, np.full(m, op[0]), np.full(m, op[1]), np.full(m, op[2]))),
        dtype=np.float64,
    )


def standardise(x: np.ndarray) -> np.ndarray:
    """Zero-mean/unit-variance columns, for the isotropic ``ScaledRBFModel`` runs."""
    return (x - x.mean(axis=0)) / x.std(axis=0)
