"""Per-cell plugin of the ``full_gp`` mode on MI355X: ``BatteryCellGP_Full`` and ``build_cellmodel_full``.

Written against the CONTRACT the reference's callers rely on, not against the reference's implementation:

* the ``IBatteryCellGP`` protocol (``src/batt_models/batt_cell_gp_protocol.py:9-86``): numpy ``[N, 4]`` inputs in
  the order (time [d], current [A], SOC [%], temperature [degC]), numpy ``[N]`` target, the seven methods below;
* the keyword names a caller may pass and read back (``src/batt_models/battcellgp_full.py:94-111``), the CSV row
  labels of a saved hyper-parameter file (``:25-32`` + "Marginal Likelihood") and the result column names
  ``t`` / ``r0_acausal_<tag>`` / ``r0var_acausal_<tag>`` (``:212-218``) - kept below as DATA TABLES;
* what ``BattGP_Full`` touches on a cell model (``src/batt_models/battgp_full.py:41-125``): ``.model`` (deletable,
  with ``train_inputs[0]`` / ``train_targets``), ``.cellnr``, ``predict_r0_op(op=, t=)``.

All GP algebra runs in libbattgp.so (HIP, gfx950) through :class:`battgp_amd.cell_gp.BatteryCellGP`; when a GPU is
visible the training tensors live on it and a prediction moves only the 4 M query doubles up and 2 M results down.
"""

from __future__ import annotations

import copy
import os
from dataclasses import dataclass
from typing import Any, Callable, Optional

import numpy as np
import pandas as pd
import torch

from . import config as cfg
from . import training
from .cell_gp import BatteryCellGP
from .engine import as_device_index


# ---- the contract, as data ---------------------------------------------------------------------------------
@dataclass(frozen=True)
class _HyperSpec:
    key: str  # keyword / params key; the model property of the same name holds the value
    labels: tuple  # CSV row label per component
    default: Any
    default_range: Any

    @property
    def range_key(self) -> str:
        return f"{self.key}_range"

    @property
    def constraint_attr(self) -> str:
        return f"{self.key}_constraint"


_HYPERS = (
    _HyperSpec("noise_variance", ("Noise Variance",), cfg.NOISE_VARIANCE, cfg.NOISE_VARIANCE_RANGE),
    _HyperSpec("outputscale_wiener", ("Wiener Outputscale",), cfg.OUTPUTSCALE_WIENER, cfg.OUTPUTSCALE_WIENER_RANGE),
    _HyperSpec("outputscale_rbf", ("RBF Outputscale",), cfg.OUTPUTSCALE_RBF, cfg.OUTPUTSCALE_RBF_RANGE),
    _HyperSpec(
        "lengthscale_rbf", ("RBF Lengthscale 1", "RBF Lengthscale 2", "RBF Lengthscale 3"), cfg.LENGTHSCALE_RBF, cfg.LENGTHSCALE_RBF_RANGE
    ),
)
# keyword -> (default, name of the trainer argument it feeds; None = not a trainer argument)
_SETTINGS = {
    "max_iter": (cfg.OPTIM_MAX_ITER, "max_iter"),
    "rel_tol": (cfg.OPTIM_REL_TOL, "rel_ftol"),
    "lr": (cfg.OPTIM_LR, "lr"),
    "dtype": (cfg.DTYPE, None),
    "n_devices": (1, None),
    "output_device": (None, None),
}
_NOT_RECORDED = frozenset({"device"})  # accepted keywords that never appear in get_parameters()
_LML_ROW = "Marginal Likelihood"
_TRAINERS: dict[str, Callable] = {
    "torch_adam": training.train_exact_gp_adam,
    "torch_lbfgs": training.train_exact_gp_lbfgs,
    "botorch_lbfgs_B": training.train_exact_gp_botorch,
}


def _series_tag(cellnr) -> str:
    """``pack`` for the pack model (cell number -1), ``c<n>`` for a cell (``src/batt_models/cellnr.py:4-8``)."""
    return "pack" if cellnr == -1 else f"c{cellnr}"


def _flat(value) -> list:
    """Components of a hyper-parameter: tuples / arrays as they are, ``(2.33e-6,)`` and ``2.33e-6`` alike."""
    return list(np.atleast_1d(np.asarray(value, dtype=object)).ravel())


def hyperparameter_table(params: dict, lml=None) -> pd.DataFrame:
    """One row per hyper-parameter component, column ``params`` - the layout of ``<cellnr>hyperparams.csv``."""
    labels, values = [], []
    for spec in _HYPERS:
        comps = _flat(params[spec.key])
        if len(spec.labels) == 1:
            comps = [params[spec.key]]  # a scalar (or the reference's 1-tuple noise) is stored as given
        for label, comp in zip(spec.labels, comps):
            labels.append(label)
            values.append(comp)
    table = pd.DataFrame({"params": pd.Series(values, index=labels, dtype=object)})
    if lml is not None:
        table.loc[_LML_ROW] = lml
    return table


def _host_vector(t: torch.Tensor) -> np.ndarray:
    return np.asarray(t.cpu(), dtype=np.float64).ravel()


def _torch_device(device) -> torch.device:
    """``device=`` as the callers pass it: ``None``, an index (``gp_runner.py:162-171``), a string or a
    ``torch.device``.  There is no CPU engine: "not given" means GPU 0 (the reference's default is the CPU), and an
    explicit CPU device fails when the first computation asks for the engine."""
    if device is None:
        return torch.device("cuda", 0)
    if isinstance(device, (int, np.integer)):
        return torch.device("cuda", int(device))
    return torch.device(device)


class BatteryCellGP_Full:
    """``IBatteryCellGP`` for one cell (or the pack): exact GP with the Wiener + ARD-RBF kernel on one MI355X."""

    def __init__(self, x, y, cellnr=None, **kwargs):
        self._cellnr = cellnr
        self.params = BatteryCellGP_Full.get_default_parameters()
        for name in kwargs:
            if name not in self.params and name not in _NOT_RECORDED:
                raise ValueError(f"unknown keyword parameter '{name}'")
        self.params.update({k: v for k, v in kwargs.items() if k not in _NOT_RECORDED})

        self.dtype_ = self.params["dtype"]
        if self.dtype_ != torch.float64:
            raise ValueError("battgp_amd computes the full_gp path in fp64 only (cfg.DTYPE)")
        self.device_ = _torch_device(kwargs.get("device"))

        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if x.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("x must be [N, 4] and y [N]")
        # the containers the callers read (train_inputs[0][:, 0].detach().cpu()) live on the GPU when there is one,
        # so the engine adopts them without a host copy
        home = self.device_ if (self.device_.type == "cuda" and torch.cuda.is_available()) else torch.device("cpu")
        self.model: BatteryCellGP = BatteryCellGP(
            torch.from_numpy(x).to(home),
            torch.from_numpy(y).to(home),
            n_devices=self.params["n_devices"],
            output_device=self.params["output_device"],
            device=self.device_,
        )
        for spec in _HYPERS:  # range first: the value is stored through the constraint's transform
            setattr(self.model, spec.constraint_attr, self.params[spec.range_key])
            setattr(self.model, spec.key, self.params[spec.key])
        # (a fresh BatteryCellGP and its likelihood are already in eval mode)

    # -- identity / parameters ------------------------------------------------------------------------------
    @property
    def cellnr(self) -> int:
        return self._cellnr

    @staticmethod
    def get_default_parameters():
        out: dict[str, Any] = {spec.key: spec.default for spec in _HYPERS}
        out.update({spec.range_key: spec.default_range for spec in _HYPERS})
        out.update({key: default for key, (default, _) in _SETTINGS.items()})
        return out

    def get_parameters(self):
        return {key: copy.deepcopy(value) for key, value in self.params.items()}

    def save_hyperparameters(self, path):
        table = hyperparameter_table(self.params, lml=self.marginallikelihood)
        table.to_csv(os.path.join(path, f"{self._cellnr}hyperparams.csv"))

    # -- training -------------------------------------------------------------------------------------------
    def train_hyperparameters(self, messages=True):
        """Optimise the hyper-parameters with the algorithm named in ``cfg.HYPER_OPT_PARAMS`` (each iteration = one
        resident re-fit + one gradient pass on the GPU), write the optimum back into ``params`` and remember the
        final loss (``-mll * N``) as ``marginallikelihood``."""
        algo = cfg.HYPER_OPT_PARAMS["opt_algorithm"]
        if algo not in _TRAINERS:
            raise ValueError(f"{algo} is not implemented as optimization algorithm.")
        trainer_args = {arg: self.params[key] for key, (_, arg) in _SETTINGS.items() if arg is not None}
        targets = self.model.train_targets
        history = _TRAINERS[algo](
            self.model, self.model.train_inputs[0], targets, loss_scale=len(targets), messages=messages, **trainer_args
        )
        for spec in _HYPERS:
            learned = _host_vector(getattr(self.model, spec.key))
            if len(spec.labels) > 1:
                self.params[spec.key] = tuple(learned)
            else:
                self.params[spec.key] = float(learned[0])
        self.marginallikelihood = history[-1] if isinstance(history, np.ndarray) else history
        return history

    # -- prediction -----------------------------------------------------------------------------------------
    def predict(self, x, full_cov=False, no_cov=False):
        """Posterior of the latent resistance at ``x [M, 4]``: ``mean`` alone (``no_cov``), ``(mean, var [M])``, or
        ``(mean, [M, M])`` with ``full_cov`` - which, like the reference, carries only the marginal variances on
        the diagonal of an otherwise-NaN matrix."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if no_cov:
            return self.model.posterior_mean(x)  # cross fill + GEMV against alpha, no triangular solve
        mean, var = map(_host_vector, self.model.posterior(x))
        if not full_cov:
            return mean, var
        square = np.full((len(var), len(var)), np.nan, dtype=np.float64)
        np.fill_diagonal(square, var)
        return mean, square

    def predict_r0_op(self, op, t: np.ndarray) -> pd.DataFrame:
        """Resistance over time at the operating point ``op`` (anything with ``.I``, ``.SOC``, ``.T``)."""
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        query = np.empty((t.size, 4), dtype=np.float64)
        query[:, 0] = t
        query[:, 1:] = (op.I, op.SOC, op.T)
        r0, r0_var = self.predict(query)
        tag = f"acausal_{_series_tag(self._cellnr)}"
        return pd.DataFrame({"t": t, f"r0_{tag}": r0, f"r0var_{tag}": r0_var})

    def get_training_data(self):
        """``(X [N, 4], y [N])`` as numpy arrays."""
        return np.asarray(self.model.train_inputs[0].cpu()), _host_vector(self.model.train_targets)

    # -- lifetime -------------------------------------------------------------------------------------------
    def __delattr__(self, name):
        # ``del cellmodel.model`` (src/batt_models/battgp_full.py:103,118) must give the engine handle and its HBM
        # back at once, without waiting for the garbage collector
        if name == "model":
            held = self.__dict__.get("model")
            if held is not None:
                held.close()
        super().__delattr__(name)


def build_cellmodel_full(cellnr, batt_data, max_training_data, max_age=None, device=None, **kwargs):
    """Factory used by ``BattGP_Full`` (``src/batt_models/battgp_full.py:41-60``): ``batt_data`` is anything with the
    reference's ``generateTrainingData(cellnr, max_training_data, max_age) -> (X[N, 4], y[N])``."""
    inputs, resistance = batt_data.generateTrainingData(cellnr, max_training_data, max_age)
    return BatteryCellGP_Full(inputs, resistance, cellnr=cellnr, device=device, **kwargs)


def predict_cells_concurrently(cellmodels, op, t: np.ndarray) -> list[pd.DataFrame]:
    """One GP per GPU, concurrently: models bound to different devices are driven
    from one thread each - the C-ABI calls release the GIL and distinct handles are independent.  Models that
    share a device run one after another on that device's thread."""
    from concurrent.futures import ThreadPoolExecutor

    lanes: dict[int, list[int]] = {}
    for i, mdl in enumerate(cellmodels):
        lanes.setdefault(as_device_index(mdl.device_), []).append(i)
    frames: list[Optional[pd.DataFrame]] = [None] * len(cellmodels)

    def drive(indices):
        for i in indices:
            frames[i] = cellmodels[i].predict_r0_op(op, t)

    with ThreadPoolExecutor(max_workers=max(1, len(lanes))) as pool:
        list(pool.map(drive, lanes.values()))
    return frames
