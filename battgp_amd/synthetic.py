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


class SyntheticBattData:
    """Stand-in for the reference's ``BattData`` with only the contract the ``full_gp`` drivers use
    (``src/batt_data/batt_data.py``): ``id``, ``age``, ``cell_nrs``, ``mean_op`` / ``median_op`` and
    ``generateTrainingData(cellnr, max_training_data, max_age) -> (X[N,4], y[N])``.  Every cell gets its
    own current / temperature / resistance realisation (cell -1 is the pack model), like the per-cell
    sensors of the field data; the data set itself is not available offline."""

    def __init__(self, batt_id: str = "synthetic", n_cells: int = 8, age_days: float = 1200.0, seed: int = 0):
        from .operating_point import Op

        self.id = batt_id
        self.age = age_days
        self.cell_nrs = list(range(1, n_cells + 1))
        self.seed = seed
        self.mean_op = Op(*REF_OP)
        self.median_op = Op(*REF_OP)

    def generateTrainingData(self, cellnr: int, max_training_data: int, max_age=None):
        age = self.age if max_age is None else min(self.age, max_age)
        x, y = make_cell_data(max_training_data, seed=self.seed * 1000 + 17 * (cellnr + 2), age_days=age)
        # cells age slightly differently: a per-cell offset and slope on top of the common trend
        y = y + 2e-4 * ((cellnr * 7919) % 13 - 6) / 6.0 + 2e-7 * ((cellnr * 104729) % 11) * x[:, 0]
        return x, y
