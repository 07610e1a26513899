"""``BatteryCellGP_Full`` / ``build_cellmodel_full`` - the per-cell plugin of the ``full_gp`` mode.

Same public surface as the reference's ``src/batt_models/battcellgp_full.py:46-240`` (an
``IBatteryCellGP``, ``src/batt_models/batt_cell_gp_protocol.py:9-86``): numpy ``[N, 4]`` / ``[N]``
in, numpy / pandas out, so ``BattGP_Full`` (``src/batt_models/battgp_full.py:41-125``),
``BattGP.train_hyperparameters`` / ``save_hyperparameters`` (``src/batt_models/battgp.py:181-225``)
and the plotting helpers consume it unchanged - with the GP algebra running on MI355X through
libbattgp.so instead of gpytorch.
"""

from __future__ import annotations

import os
from copy import deepcopy
from typing import Any, Optional

import numpy as np
import pandas as pd
import torch

from . import config as cfg
from . import training
from .cell_gp import BatteryCellGP
from .engine import as_device_index
from .operating_point import Op, get_causal_tag, get_cell_tag


def _create_hyperparams_df(params: dict) -> pd.DataFrame:
    """Row labels and order of ``battcellgp_full.py:21-43``."""
    return pd.DataFrame(
        index=[
            "Noise Variance",
            "Wiener Outputscale",
            "RBF Outputscale",
            "RBF Lengthscale 1",
            "RBF Lengthscale 2",
            "RBF Lengthscale 3",
        ],
        columns=["params"],
        data=[
            [params["noise_variance"]],
            [params["outputscale_wiener"]],
            [params["outputscale_rbf"]],
            [params["lengthscale_rbf"][0]],
            [params["lengthscale_rbf"][1]],
            [params["lengthscale_rbf"][2]],
        ],
    )


def _resolve_device(device) -> torch.device:
    """The reference defaults to CPU; this engine has no CPU path, so "not given" means GPU 0 and an
    explicit CPU device is an error raised when the first computation needs the engine."""
    if device is None:
        return torch.device("cuda", 0)
    if isinstance(device, (int, np.integer)):
        return torch.device("cuda", int(device))
    return torch.device(device)


class BatteryCellGP_Full:
    def __init__(self, x: np.ndarray, y: np.ndarray, cellnr: Optional[int] = None, **kwargs):
        self.params: dict[str, Any] = self.get_default_parameters()
        self._cellnr = cellnr

        UNLOGGED_PARAMS = {"device"}
        for k, v in kwargs.items():
            if k in UNLOGGED_PARAMS:
                continue
            if k not in self.params:
                raise ValueError(f"unknown keyword parameter '{k}'")
            self.params[k] = v

        self.dtype_ = self.params["dtype"]
        if self.dtype_ != torch.float64:
            raise ValueError("battgp_amd computes the full_gp path in fp64 only (cfg.DTYPE)")
        self.device_ = _resolve_device(kwargs.get("device", None))

        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if x.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("x must be [N, 4] and y [N]")
        # torch tensors are the containers the callers expect (train_inputs[0][:, 0], .detach().cpu());
        # they live on the GPU when one is visible so the fit can adopt them without a host copy
        on_gpu = self.device_.type == "cuda" and torch.cuda.is_available()
        tdev = self.device_ if on_gpu else torch.device("cpu")
        xt = torch.tensor(x, dtype=self.dtype_, device=tdev)
        yt = torch.tensor(y, dtype=self.dtype_, device=tdev)

        self.model: BatteryCellGP = BatteryCellGP(
            xt,
            yt,
            n_devices=self.params["n_devices"],
            output_device=self.params["output_device"],
            device=self.device_,
        )
        self.model.noise_variance_constraint = self.params["noise_variance_range"]
        self.model.noise_variance = self.params["noise_variance"]
        self.model.outputscale_wiener_constraint = self.params["outputscale_wiener_range"]
        self.model.outputscale_wiener = self.params["outputscale_wiener"]
        self.model.outputscale_rbf_constraint = self.params["outputscale_rbf_range"]
        self.model.outputscale_rbf = self.params["outputscale_rbf"]
        self.model.lengthscale_rbf_constraint = self.params["lengthscale_rbf_range"]
        self.model.lengthscale_rbf = self.params["lengthscale_rbf"]
        self.model.eval()
        self.model.likelihood.eval()

    @property
    def cellnr(self) -> int:
        return self._cellnr

    def __delattr__(self, name):
        # ``del cellmodel.model`` (src/batt_models/battgp_full.py:103,118) must release the engine handle and
        # its HBM at once, without waiting for the garbage collector
        if name == "model":
            mdl = self.__dict__.get("model")
            if mdl is not None:
                mdl.close()
        super().__delattr__(name)

    @staticmethod
    def get_default_parameters() -> dict[str, Any]:
        return {
            "noise_variance": cfg.NOISE_VARIANCE,
            "outputscale_wiener": cfg.OUTPUTSCALE_WIENER,
            "outputscale_rbf": cfg.OUTPUTSCALE_RBF,
            "lengthscale_rbf": cfg.LENGTHSCALE_RBF,
            "noise_variance_range": cfg.NOISE_VARIANCE_RANGE,
            "outputscale_wiener_range": cfg.OUTPUTSCALE_WIENER_RANGE,
            "outputscale_rbf_range": cfg.OUTPUTSCALE_RBF_RANGE,
            "lengthscale_rbf_range": cfg.LENGTHSCALE_RBF_RANGE,
            "max_iter": cfg.OPTIM_MAX_ITER,
            "rel_tol": cfg.OPTIM_REL_TOL,
            "lr": cfg.OPTIM_LR,
            "dtype": cfg.DTYPE,
            "n_devices": 1,
            "output_device": None,
        }

    def get_parameters(self) -> dict[str, Any]:
        return deepcopy(self.params)

    def save_hyperparameters(self, path: str) -> None:
        hyperparams_df = _create_hyperparams_df(self.params)
        hyperparams_df.loc["Marginal Likelihood"] = self.marginallikelihood
        path = os.path.join(path, f"{self._cellnr}hyperparams.csv")
        hyperparams_df.to_csv(path)

    def train_hyperparameters(self, messages: bool = True) -> np.ndarray:
        x_train = self.model.train_inputs[0]
        y_train = self.model.train_targets

        algo = cfg.HYPER_OPT_PARAMS["opt_algorithm"]
        if algo == "torch_lbfgs":
            trainer = training.train_exact_gp_lbfgs
        elif algo == "botorch_lbfgs_B":
            trainer = training.train_exact_gp_botorch
        elif algo == "torch_adam":
            trainer = training.train_exact_gp_adam
        else:
            raise ValueError(f"{algo} is not implemented as optimization algorithm.")

        losses = trainer(
            self.model,
            x_train,
            y_train,
            loss_scale=len(y_train),
            max_iter=self.params["max_iter"],
            rel_ftol=self.params["rel_tol"],
            lr=self.params["lr"],
            messages=messages,
        )

        self.params["noise_variance"] = float(self.model.noise_variance)
        self.params["outputscale_wiener"] = float(self.model.outputscale_wiener)
        self.params["outputscale_rbf"] = float(self.model.outputscale_rbf)
        self.params["lengthscale_rbf"] = tuple(self.model.lengthscale_rbf.detach().cpu().numpy()[0])

        if isinstance(losses, np.ndarray):
            self.marginallikelihood = losses[-1]
        else:
            self.marginallikelihood = losses
        return losses

    def predict(self, x: np.ndarray, full_cov: bool = False, no_cov: bool = False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if no_cov:
            # mean only: cross fill + GEMV against the cached alpha, no triangular solve
            return self.model.posterior_mean(x)
        out = self.model(torch.as_tensor(x))
        y = out.mean.detach().cpu().numpy()
        y_var = out.variance.detach().cpu().numpy().reshape(-1)
        if full_cov:
            # the reference returns a NaN matrix with only the diagonal filled (battcellgp_full.py:182-193)
            n = x.shape[0]
            covmatrix = np.full((n, n), np.nan, dtype=np.float64)
            covmatrix[np.diag_indices(n)] = y_var
            y_var = covmatrix
        return (y, y_var)

    def predict_r0_op(self, op: Op, t: np.ndarray) -> pd.DataFrame:
        t = np.asarray(t, dtype=np.float64)
        X = np.column_stack((t, np.ones(len(t)) * op.I, np.ones(len(t)) * op.SOC, np.ones(len(t)) * op.T))
        (r0, r0var) = self.predict(X, full_cov=False)
        cell_tag = get_cell_tag(self._cellnr)
        causal_tag = get_causal_tag(False)
        return pd.DataFrame(
            {
                "t": t,
                f"r0_{causal_tag}_{cell_tag}": r0,
                f"r0var_{causal_tag}_{cell_tag}": r0var,
            }
        )

    def get_training_data(self) -> tuple[np.ndarray, np.ndarray]:
        return (
            self.model.train_inputs[0].detach().cpu().numpy(),
            self.model.train_targets.detach().cpu().numpy().reshape((-1,)),
        )


def build_cellmodel_full(
    cellnr: int,
    batt_data,
    max_training_data: int,
    max_age: Optional[int] = None,
    device=None,
    **kwargs,
) -> BatteryCellGP_Full:
    """``src/batt_models/battcellgp_full.py:229-240``: ``batt_data`` is anything with the
    reference's ``generateTrainingData(cellnr, max_training_data, max_age) -> (X[N,4], y[N])``."""
    (x, y) = batt_data.generateTrainingData(cellnr, max_training_data, max_age)
    return BatteryCellGP_Full(x, y, cellnr, device=device, **kwargs)


def predict_cells_concurrently(cellmodels, op: Op, t: np.ndarray) -> list[pd.DataFrame]:
    """One GP per GPU, concurrently: models bound to different devices
    are driven from one thread each - the C-ABI calls release the GIL and distinct handles are
    independent.  Models that share a device run one after another on that device's thread."""
    from concurrent.futures import ThreadPoolExecutor

    by_dev: dict[int, list[int]] = {}
    for i, mdl in enumerate(cellmodels):
        by_dev.setdefault(as_device_index(mdl.device_), []).append(i)
    out: list[Optional[pd.DataFrame]] = [None] * len(cellmodels)

    def run(idxs):
        for i in idxs:
            out[i] = cellmodels[i].predict_r0_op(op, t)

    with ThreadPoolExecutor(max_workers=max(1, len(by_dev))) as pool:
        list(pool.map(run, by_dev.values()))
    return out
