"""System-level ``full_gp`` driver: one pack model + one model per cell, all predicted on a common
300-point time grid at the reference operating point, merged into one data frame and saved as
feather + json.  Same public behaviour as the reference's ``BattGP_Full`` / ``BattGP`` / ``BattGPResult``
(``src/batt_models/battgp_full.py:15-125``, ``src/batt_models/battgp.py:16-262``) - so the
``gp_runner.py`` ``full_gp`` branch (``:68-96``) and ``calc_fault_probabilities`` consume it unchanged -
with MI355X-native additions: the 1 + n_cells GPs of a system are independent, so at the reference's sizes
(N = 1000 ... 16 000 per cell, where one GP is a latency-bound chain of small kernels using a fraction of the 256 CUs)
they are driven CONCURRENTLY on the one GPU - one engine handle and stream set each, one host thread each, the
C-ABI calls release the GIL (``in_flight=``; automatic: as many as fit into half of the free HBM); ``devices=``
spreads them over several GPUs; and models are freed right after use because at the large sizes one N x N fp64
factor fills a card.  Every GP runs the same kernels on its own buffers whatever the concurrency: the numbers
do not depend on it.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Literal, Optional, Union

import numpy as np
import pandas as pd

from .battcellgp_full import BatteryCellGP_Full, build_cellmodel_full
from .operating_point import Op, get_causal_tag, get_cell_tag


class RefStrategy:
    """Which operating point the system is evaluated at: ``RefStrategy("mean")``, ``RefStrategy("median")`` (of the
    battery's data) or ``RefStrategy(Op(...))`` - the argument ``gp_runner.py:53-76`` hands to ``BattGP_Full``
    (``src/batt_models/ref_strategy.py:7-47``: same constructor, same queries, ``ValueError`` for anything else)."""

    def __init__(self, strategy="mean"):
        self._value = None
        if isinstance(strategy, str):
            if strategy not in ("mean", "median"):
                raise ValueError("Invalid value for 'strategy'")
            self._kind = strategy
        elif all(hasattr(strategy, name) for name in ("I", "SOC", "T")):  # an Op (this package's or the reference's)
            self._kind = "manual"
            self._value = strategy
        else:
            raise ValueError("Invalid value for 'strategy'")

    def is_mean(self) -> bool:
        return self._kind == "mean"

    def is_median(self) -> bool:
        return self._kind == "median"

    def is_manual(self) -> bool:
        return self._kind == "manual"

    def get_manual_value(self):
        if self._kind != "manual":
            raise ValueError("manual value can only be given if strategy is 'manual'")
        return self._value


def _resolve_ref_op(batt_data, ref_strategy, ref_op):
    """The operating point a ``BattGP`` starts with (``src/batt_models/battgp.py:115-121``).  ``ref_strategy`` may be
    the reference's own ``RefStrategy`` object (anything with ``is_median`` / ``is_mean`` / ``get_manual_value``), this
    module's, a bare ``"mean"`` / ``"median"`` or an ``Op``; ``ref_op=`` (an addition of this package) wins."""
    if ref_op is not None:
        return ref_op
    if not all(hasattr(ref_strategy, name) for name in ("is_median", "is_mean", "get_manual_value")):
        ref_strategy = RefStrategy(ref_strategy)
    if ref_strategy.is_median():
        return batt_data.median_op
    if ref_strategy.is_mean():
        return batt_data.mean_op
    return ref_strategy.get_manual_value()


@dataclass
class BattGPResult:
    """``src/batt_models/battgp.py:16-92``."""

    batt_data: Any
    cellmodels: List[BatteryCellGP_Full]
    ref_op: Op
    df: pd.DataFrame

    def get_cell_data(
        self,
        cellnrs: Union[Iterable[int], int],
        signals: Optional[Iterable[str]] = None,
        causal: bool = False,
        missing_behaviour: Literal["error", "ignore"] = "error",
    ) -> pd.DataFrame:
        per_cell = ("r0", "r0var", "dr0", "dr0var")
        if signals is None:
            signals = ["t", "ds_count", "r0", "r0var", "dr0", "dr0var"]
        single = isinstance(cellnrs, (int, np.integer))
        cells = [int(cellnrs)] if single else list(cellnrs)
        ctag = get_causal_tag(causal)
        wanted: list[str] = []
        rename: dict[str, str] = {}
        for sig in signals:
            if sig not in per_cell:
                wanted.append(sig)
                rename[sig] = sig
                continue
            for c in cells:
                src = f"{sig}_{ctag}_{get_cell_tag(c)}"
                wanted.append(src)
                rename[src] = sig if single else f"{sig}_{get_cell_tag(c)}"
        present = [w for w in wanted if w in self.df.columns]
        if len(present) < len(wanted) and missing_behaviour == "error":
            missing = sorted(set(wanted) - set(present))
            raise ValueError(f"signal(s) {', '.join(missing)} not available in the result")
        return self.df[present].rename(columns=rename)


class BattGP_Full:
    def __init__(
        self,
        batt_data,
        *,
        max_training_data: Optional[int] = None,
        max_age: Optional[int] = None,
        ref_op: Optional[Op] = None,
        ref_strategy="mean",
        device=None,
        devices: Optional[list] = None,
        save_path: Optional[str] = None,
        in_flight: Optional[int] = None,
        **kwargs,
    ) -> None:
        if max_training_data is None:
            max_training_data = 2000  # battgp_full.py:27-31
            print(f"Max training data set to {max_training_data}, because no values was passed for max_training_data")
        self.batt_data = batt_data
        self.max_training_data = max_training_data
        self.max_age = batt_data.age if max_age is None else max_age
        self.ref_strategy = ref_strategy
        self.ref_op = _resolve_ref_op(batt_data, ref_strategy, ref_op)
        print(f"Reference operating point: {self.ref_op}")  # battgp.py:122
        self.save_path = None
        if save_path is not None:
            self.save_path = os.path.join(save_path, batt_data.id)
            os.makedirs(self.save_path, exist_ok=True)
        # one device for everything (the reference), or a list to deal the 1 + n_cells GPs over
        self.devices = list(devices) if devices else [device]
        # GPs in flight per device: None = automatic (see _in_flight), 1 = the reference's strictly sequential loop
        self.in_flight = in_flight
        cells = [-1] + list(batt_data.cell_nrs)
        models = [
            build_cellmodel_full(
                c, batt_data, max_training_data=max_training_data, max_age=self.max_age, device=self.devices[i % len(self.devices)], **kwargs
            )
            for i, c in enumerate(cells)
        ]
        self.packmodel: BatteryCellGP_Full = models[0]
        self.cellmodels: List[BatteryCellGP_Full] = models[1:]
        self.t = None

    # -- small accessors of the base class (battgp.py:131-179) -------------------------------------
    def set_operating_point_to_mean(self) -> None:
        op = self.batt_data.mean_op
        print(f"Battery operating point set to mean: {op.disp_str()}")
        self.set_operating_point(op, verbose=False)

    def set_operating_point_to_median(self) -> None:
        op = self.batt_data.median_op
        print(f"Battery operating point set to median: {op.disp_str()}")
        self.set_operating_point(op, verbose=False)

    def set_operating_point(self, op: Op, verbose: bool = True) -> None:
        if verbose:
            print(f"Battery operating point set to: {op.disp_str()}")
        self.ref_op = op

    def get_operating_point(self) -> Op:
        return self.ref_op

    def get_cell_model(self, cellnr: int) -> BatteryCellGP_Full:
        if cellnr == -1:
            return self.packmodel
        for cell in self.cellmodels:
            if cell.cellnr == cellnr:
                return cell
        raise ValueError(f"cell {cellnr} does not exist")

    def get_parameters(self) -> Dict[str, Any]:
        return {  # battgp_full.py:62-68
            "ref_point": self.get_operating_point().disp_str(),
            "segment_criteria": "Saving segment criteria not implemented yet, see config.py",
            "gap_removal": "Saving gap removal not implemented yet, see config.py",
            "ocv_path": "Saving ocv path not implemented yet, see config.py",
        }

    # -- hyper-parameters (battgp.py:181-231) -------------------------------------------------------
    def train_hyperparameters(self, parallelize: bool = False, messages: bool = True) -> None:
        models = [self.packmodel, *self.cellmodels]
        if not parallelize:
            for mdl in models:
                mdl.train_hyperparameters(messages=messages)
            return
        # one thread per model (handles are independent; the C-ABI calls release the GIL)
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(len(models)) as pool:
            list(pool.map(lambda mdl: mdl.train_hyperparameters(messages=False), models))

    def save_hyperparameters(self, path: str) -> None:
        save_path = os.path.join(path, self.batt_data.id)
        os.makedirs(save_path, exist_ok=True)
        for mdl in [self.packmodel, *self.cellmodels]:
            mdl.save_hyperparameters(save_path)

    # -- the hot call (battgp_full.py:70-125) -------------------------------------------------------
    def _time_grid(self, add_time_steps: bool) -> np.ndarray:
        t_train = self.cellmodels[0].model.train_inputs[0][:, 0].detach().cpu().numpy()
        if add_time_steps:
            extra = np.linspace(t_train[0], self.batt_data.age, int(self.batt_data.age - t_train[0]))
            return np.sort(np.concatenate((t_train, extra)))
        return np.linspace(t_train[0], self.batt_data.age, 300)

    @staticmethod
    def _free(model: BatteryCellGP_Full) -> None:
        # releases the engine handle (parked in the library's pool) - deterministically, through
        # BatteryCellGP_Full.__delattr__; the reference's gc.collect() + empty_cache() (battgp_full.py:104-105)
        # would cost 34 ms per cell here and is not needed
        del model.model

    def _in_flight(self, models, n_query: int, listed: int = 1) -> int:
        """How many GPs of one device run at the same time.  ``in_flight=`` decides when given; a device listed ``c``
        times in ``devices=`` asks for ``c``; otherwise automatic: up to ``AUTO_IN_FLIGHT_MAX_N`` points per GP the
        factorisation is a latency-bound chain that leaves most of the chip idle, so as many GPs as fit into half of
        the device's free memory (each holds its factor ``8 (N + 64 + M) N`` bytes plus panel workspaces) run
        together; larger GPs fill the chip on their own and run one after another."""
        if self.in_flight is not None:
            return max(1, min(int(self.in_flight), len(models)))
        if listed > 1:
            return min(listed, len(models))
        n = max(int(m.model.train_targets.shape[0]) for m in models)
        if n > self.AUTO_IN_FLIGHT_MAX_N:
            return 1
        per_gp = 8.0 * (n + 64 + n_query) * n * 1.1 + 64e6
        try:
            import torch

            free, _total = torch.cuda.mem_get_info(models[0].device_)
        except Exception:  # no GPU visible: the engine will say so on the first call
            return 1
        return max(1, min(len(models), int(0.5 * free / per_gp)))

    AUTO_IN_FLIGHT_MAX_N = 20000

    def predict_cell_r0_op(self, destroy_after_run: bool = True, add_time_steps: bool = False, save: bool = True) -> BattGPResult:
        self.t = self._time_grid(add_time_steps)
        models = [self.packmodel, *self.cellmodels]
        frames: list[Optional[pd.DataFrame]] = [None] * len(models)
        keep = len(models) - 1  # the last cell model stays alive for the plots (plotting.py:233,259)

        def run_one(i):
            frames[i] = models[i].predict_r0_op(op=self.ref_op, t=self.t)
            if destroy_after_run and i != keep:
                self._free(models[i])

        # the constructor bound model i to devices[i % n]; the models of one (distinct) device share a queue served by
        # that device's worker threads
        by_device: dict = {}
        for i in range(len(models)):
            by_device.setdefault(str(self.devices[i % len(self.devices)]), []).append(i)
        lanes = list(by_device.values())
        listed = [sum(str(d) == key for d in self.devices) for key in by_device]
        workers = [self._in_flight([models[i] for i in lane], len(self.t), c) for lane, c in zip(lanes, listed)]
        if sum(workers) <= 1:
            for i in range(len(models)):
                run_one(i)
        else:
            from concurrent.futures import ThreadPoolExecutor

            lane_pools = [ThreadPoolExecutor(w) for w in workers]
            try:
                futures = [lp.submit(run_one, i) for lp, lane in zip(lane_pools, lanes) for i in lane]
                for f in futures:
                    f.result()  # re-raises a worker's exception (NotPSDError, EngineError) in the caller
            finally:
                for lp in lane_pools:
                    lp.shutdown(wait=True)
        # every frame carries the same time grid self.t, so the reference's chain of DataFrame.merge() calls on
        # "t" (battgp_full.py:100-121; ~1 ms each) is a column-wise concatenation - as long as the grid has no
        # repeated time stamps: on repeated keys merge() yields their cross product (add_time_steps=True repeats
        # t_train[0], the start of the linspace), and the callers downstream get exactly that frame here too
        unique_grid = np.unique(self.t).size == self.t.size
        if unique_grid and all(np.array_equal(f["t"].to_numpy(), frames[0]["t"].to_numpy()) for f in frames[1:]):
            df = pd.concat([frames[0]] + [f.drop(columns="t") for f in frames[1:]], axis=1)
        else:
            df = frames[0]
            for f in frames[1:]:
                df = df.merge(f)
        if save and self.save_path is not None:
            self.save_df(df)
        return BattGPResult(self.batt_data, self.cellmodels, self.ref_op, df)

    def save_df(self, df: Optional[pd.DataFrame]) -> None:
        """feather + json pair, data file removed first so the two never disagree (battgp.py:233-262)."""
        f_data = os.path.join(self.save_path, "battgpf_df.feather")
        f_info = os.path.join(self.save_path, "battgpf_info.json")
        if df is not None:
            for f in (f_data, f_info):
                try:
                    os.remove(f)
                except FileNotFoundError:
                    pass
            df.to_feather(f_data)
        with open(f_info, "w") as fil:
            json.dump(self.get_parameters(), fil)
