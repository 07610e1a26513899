"""``BatteryCellGP`` - the exact-GP *model object* of the ``full_gp`` path, MI355X-native.

Duck-types what the reference's callers touch on ``gpytorch.models.ExactGP`` as specialised in
``src/batt_models/cell_gp.py:12-194``:

* zero mean, ``Scale(Wiener[t]) + Scale(ARD-RBF[I, SOC, T])`` + Gaussian noise (``:27-36``);
* ``train_inputs`` (tuple holding the ``[N, 4]`` tensor), ``train_targets`` (``[N]``) - read by
  ``BattGP_Full.predict_cell_r0_op`` (``src/batt_models/battgp_full.py:84,98``) and by
  ``BatteryCellGP_Full.get_training_data`` (``battcellgp_full.py:222-226``);
* the hyper-parameter / constraint properties (``:64-194``);
* ``train()`` / ``eval()`` / ``likelihood`` / ``parameters()`` and ``model(x) -> .mean, .variance``
  (``battcellgp_full.py:171-180``).

Everything numeric happens in libbattgp.so (HIP); torch tensors are containers only.  The engine
handle is created lazily on first use and freed when the object is deleted - the reference frees
GPU memory the same way (``del cellmodel.model``, ``battgp_full.py:102-120``).

Differences from the reference, on purpose (findings 6 and 9):
hyper-parameters are used verbatim in fp64 (the reference rounds them through fp32 raw parameters),
and the factorisation is always the exact jittered Cholesky (GPyTorch switches to CG/Lanczos above
N = 800 by default).
"""

from __future__ import annotations

import weakref

import math
from typing import Iterable

import numpy as np
import torch

from . import KERNEL_BATTGP
from .engine import ExactGPEngine, as_device_index

MIN_VARIANCE = 1e-10  # gpytorch.settings.min_variance for fp64 (MultivariateNormal.variance)


class Constraint:
    """``gpytorch.constraints.Interval/Positive/GreaterThan/LessThan`` stand-in with the same
    raw <-> value transforms (sigmoid on intervals, softplus on half-lines); built by
    :func:`constraint_from_range` exactly as ``src/gpytorch_utils.py:17-33,56-80`` chooses them."""

    def __init__(self, lower, upper):
        self.lower_bound = np.asarray(lower, dtype=np.float64)
        self.upper_bound = np.asarray(upper, dtype=np.float64)

    @property
    def kind(self) -> str:
        lo_inf = np.all(np.isneginf(self.lower_bound))
        hi_inf = np.all(np.isposinf(self.upper_bound))
        if lo_inf and hi_inf:
            return "free"
        if hi_inf:
            return "greater"
        if lo_inf:
            return "less"
        return "interval"

    def transform(self, raw):
        raw = np.asarray(raw, dtype=np.float64)
        k = self.kind
        if k == "free":
            return raw
        if k == "greater":
            return self.lower_bound + np.logaddexp(0.0, raw)
        if k == "less":
            return self.upper_bound - np.logaddexp(0.0, -raw)
        return self.lower_bound + (self.upper_bound - self.lower_bound) / (1.0 + np.exp(-raw))

    def inverse_transform(self, value):
        value = np.asarray(value, dtype=np.float64)
        k = self.kind
        if k == "free":
            return value
        if k == "greater":
            d = value - self.lower_bound
            return d + np.log(-np.expm1(-d))
        if k == "less":
            d = self.upper_bound - value
            return -(d + np.log(-np.expm1(-d)))
        p = (value - self.lower_bound) / (self.upper_bound - self.lower_bound)
        return np.log(p) - np.log1p(-p)

    def dvalue_draw(self, raw):
        raw = np.asarray(raw, dtype=np.float64)
        k = self.kind
        if k == "free":
            return np.ones_like(raw)
        sig = 1.0 / (1.0 + np.exp(-raw))
        if k == "greater":
            return sig
        if k == "less":
            return 1.0 - sig
        return (self.upper_bound - self.lower_bound) * sig * (1.0 - sig)

    def to(self, _device):
        return self

    def __repr__(self):
        return f"Constraint({self.kind}, {self.lower_bound}, {self.upper_bound})"


def constraint_from_range(value) -> Constraint:
    """``(lo, hi)`` or iterable of ``(lo, hi)`` -> :class:`Constraint`
    (``src/gpytorch_utils.py:17-33`` scalar, ``:56-80`` vector)."""
    if isinstance(value, Constraint):
        return value
    first = value[0]
    if isinstance(first, (tuple, list, np.ndarray)):
        lo = [float(v[0]) for v in value]
        hi = [float(v[1]) for v in value]
        return Constraint(lo, hi)
    return Constraint(float(value[0]), float(value[1]))


def _as_float_array(values) -> np.ndarray:
    """``get_tensor`` of the reference (``src/gpytorch_utils.py:36-53``) but in fp64."""
    if isinstance(values, torch.Tensor):
        return values.detach().cpu().double().numpy().reshape(-1)
    if isinstance(values, Iterable):
        return np.asarray([float(v) for v in values], dtype=np.float64)
    return np.asarray([float(values)], dtype=np.float64)


class _Likelihood:
    """The slice of ``GaussianLikelihood`` the callers use."""

    def __init__(self, owner: "BatteryCellGP"):
        # weak back-reference: no reference cycle, so ``del cellmodel.model`` releases the engine handle by
        # reference counting alone (the reference needs gc.collect() for that, 34 ms per call here)
        self._owner = weakref.proxy(owner)
        self.training = False

    @property
    def noise(self) -> torch.Tensor:
        return torch.tensor([self._owner._noise], dtype=torch.float64)

    @noise.setter
    def noise(self, value):
        self._owner._noise = float(_as_float_array(value)[0])
        self._owner._invalidate()

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, _device):
        return self


class Posterior:
    """Result of ``model(x)``: ``.mean`` and ``.variance`` as torch tensors, like
    ``gpytorch.distributions.MultivariateNormal`` (``battcellgp_full.py:173-180``)."""

    def __init__(self, mean: np.ndarray, variance: np.ndarray, device):
        self.mean = torch.as_tensor(mean, dtype=torch.float64, device=device)
        self.variance = torch.as_tensor(variance, dtype=torch.float64, device=device)


class BatteryCellGP:
    def __init__(
        self,
        train_x: torch.Tensor,
        train_y: torch.Tensor,
        *,
        n_devices: int = 1,
        output_device=None,
        device=None,
        **_kwargs,
    ):
        if n_devices < 1:
            raise ValueError("n_devices must be an integer and >= 1")  # cell_gp.py:47
        self.n_devices = int(n_devices)
        self.output_device = output_device
        # n_devices > 1 in the reference = ONE GP whose kernel matrix is spread over several GPUs
        # (MultiDeviceKernel, cell_gp.py:37-43).  Here that is the sharded Cholesky (battgp_amd/sharded.py), which
        # runs one process per GPU: the caller must already be inside a torch.distributed group of that size.
        self._sharded = None
        if self.n_devices > 1:
            self._require_process_group()
        x = torch.as_tensor(train_x, dtype=torch.float64)
        y = torch.as_tensor(train_y, dtype=torch.float64).reshape(-1)
        if x.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("train_x must be [N, D] and train_y [N]")
        self.device_ = device if device is not None else x.device
        self._resident = False  # X, y of THIS object are in the engine's HBM (cleared when they are replaced)
        self._train_inputs = (x.contiguous(),)
        self._train_targets = y.contiguous()
        d = x.shape[1]
        # GPyTorch defaults before the adaptor overwrites them (softplus(0) = ln 2)
        self._noise = math.log(2.0)
        self._outputscale_wiener = math.log(2.0)
        self._outputscale_rbf = math.log(2.0)
        self._lengthscale_rbf = np.full(d - 1, math.log(2.0))
        self._c_noise = Constraint(1e-4, np.inf)
        self._c_sw = Constraint(0.0, np.inf)
        self._c_sr = Constraint(0.0, np.inf)
        self._c_ls = Constraint(np.zeros(d - 1), np.full(d - 1, np.inf))
        self.likelihood = _Likelihood(self)
        self.training = False
        self._engine: ExactGPEngine | None = None
        self._fitted = False
        self.lml = None
        self.jitter = None

    # -- training data (what the callers read; replacing it invalidates the copy in HBM) --------------------
    @property
    def train_inputs(self):
        return self._train_inputs

    @train_inputs.setter
    def train_inputs(self, value):
        value = value if isinstance(value, (tuple, list)) else (value,)
        self._train_inputs = tuple(torch.as_tensor(v, dtype=torch.float64).contiguous() for v in value)
        self._resident = False
        self._invalidate()

    @property
    def train_targets(self):
        return self._train_targets

    @train_targets.setter
    def train_targets(self, value):
        self._train_targets = torch.as_tensor(value, dtype=torch.float64).reshape(-1).contiguous()
        self._resident = False
        self._invalidate()

    def set_train_data(self, inputs=None, targets=None, strict: bool = True):
        """``ExactGP.set_train_data``: swap the data, keep the hyper-parameters."""
        if inputs is not None:
            self.train_inputs = inputs
        if targets is not None:
            self.train_targets = targets

    def _require_process_group(self):
        import torch.distributed as dist

        from .engine import EngineError

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != self.n_devices:
            raise EngineError(
                f"n_devices={self.n_devices} spreads ONE GP over {self.n_devices} GPUs: in battgp_amd that is the sharded "
                f"Cholesky, one process per GPU - launch the caller with `python -m torch.distributed.run "
                f"--nproc-per-node {self.n_devices} ...` (battgp_amd/sharded.py); n_devices=1 is the single-GPU engine"
            )
        return dist

    # -- module plumbing ------------------------------------------------------------------------
    def to(self, device):
        self.device_ = device
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def hyp_vector(self) -> np.ndarray:
        """``[noise, s_wiener, s_rbf, l_1..l_{D-1}]`` - the C-ABI layout (include/battgp.h)."""
        return np.concatenate(
            ([self._noise, self._outputscale_wiener, self._outputscale_rbf], self._lengthscale_rbf)
        ).astype(np.float64)

    def set_hyp_vector(self, hyp) -> None:
        hyp = np.asarray(hyp, dtype=np.float64).reshape(-1)
        self._noise, self._outputscale_wiener, self._outputscale_rbf = (float(v) for v in hyp[:3])
        self._lengthscale_rbf = hyp[3:].copy()
        self._invalidate()

    def parameters(self):
        """Raw (unconstrained) parameters as fp64 tensors, in GPyTorch's registration order
        (noise, wiener outputscale, rbf outputscale, rbf lengthscale)."""
        raw = self.raw_vector()
        return [torch.tensor(raw[0:1]), torch.tensor(raw[1:2]), torch.tensor(raw[2:3]), torch.tensor(raw[3:])]

    def constraints(self):
        return [self._c_noise, self._c_sw, self._c_sr, self._c_ls]

    def raw_vector(self) -> np.ndarray:
        # after set_raw_vector() the raw values themselves are kept: value -> raw is lossy once a parameter has
        # saturated at a bound of its interval (sigmoid(raw) rounds to 0 or 1), and gpytorch stores the raw tensors
        if getattr(self, "_raw", None) is not None:
            return self._raw.copy()
        return np.concatenate(
            (
                np.atleast_1d(self._c_noise.inverse_transform(self._noise)),
                np.atleast_1d(self._c_sw.inverse_transform(self._outputscale_wiener)),
                np.atleast_1d(self._c_sr.inverse_transform(self._outputscale_rbf)),
                np.atleast_1d(self._c_ls.inverse_transform(self._lengthscale_rbf)),
            )
        )

    def set_raw_vector(self, raw) -> None:
        raw = np.asarray(raw, dtype=np.float64).reshape(-1)
        if getattr(self, "_raw", None) is not None and np.array_equal(raw, self._raw):
            return  # the optimiser came back to the point it last evaluated: the factor and the LML stay valid
        self._noise = float(self._c_noise.transform(raw[0]))
        self._outputscale_wiener = float(self._c_sw.transform(raw[1]))
        self._outputscale_rbf = float(self._c_sr.transform(raw[2]))
        self._lengthscale_rbf = np.atleast_1d(self._c_ls.transform(raw[3:])).astype(np.float64)
        self._invalidate()
        self._raw = raw.copy()

    def dvalue_draw(self) -> np.ndarray:
        raw = self.raw_vector()
        return np.concatenate(
            (
                np.atleast_1d(self._c_noise.dvalue_draw(raw[0])),
                np.atleast_1d(self._c_sw.dvalue_draw(raw[1])),
                np.atleast_1d(self._c_sr.dvalue_draw(raw[2])),
                np.atleast_1d(self._c_ls.dvalue_draw(raw[3:])),
            )
        )

    # -- engine --------------------------------------------------------------------------------
    def _invalidate(self):
        self._fitted = False
        self._raw = None  # a value or constraint was set directly: raw parameters follow from the values again

    def engine(self) -> ExactGPEngine:
        if self._engine is None:
            self._engine = ExactGPEngine(KERNEL_BATTGP, self.hyp_vector(), device=as_device_index(self.device_))
        return self._engine

    def _train_on_engine_device(self, eng: ExactGPEngine) -> bool:
        x = self._train_inputs[0]
        return bool(x.is_cuda and x.device.index == eng.device_index and self._train_targets.is_cuda)

    def _shard(self):
        """The sharded engine of an ``n_devices > 1`` model (one rank of it)."""
        if self._sharded is None:
            from .sharded import make_sharded_gp

            dist = self._require_process_group()
            self._sharded = make_sharded_gp(KERNEL_BATTGP, self.hyp_vector(), backend_name=dist.get_backend())
        return self._sharded

    def fit(self) -> float:
        """Fill + jittered Cholesky + LML on the GPU (cached until a hyper-parameter or the data changes) - the
        work GPyTorch does lazily inside the first ``model(x)`` / ``mll`` call."""
        if self._fitted:
            return self.lml
        x, y = self._train_inputs[0], self._train_targets
        if self.n_devices > 1:
            gp = self._shard()
            gp.set_hyp(self.hyp_vector())
            self.lml = gp.fit(x.detach().cpu().numpy(), y.detach().cpu().numpy())
            self.jitter = gp.jitter
            self._fitted = True
            return self.lml
        eng = self.engine()
        try:
            if self._resident:
                self.lml = eng.refit(self.hyp_vector())  # X, y of this object already in HBM
            else:
                eng.set_hyp(self.hyp_vector())
                if self._train_on_engine_device(eng):
                    torch.cuda.current_stream(x.device).synchronize()
                    self.lml = eng.fit_device(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1])
                else:
                    self.lml = eng.fit(x.detach().cpu().numpy(), y.detach().cpu().numpy())
        except Exception:
            self._resident = False  # a failed call may have dropped the resident problem: upload again next time
            raise
        self._resident = True
        self.jitter = eng.jitter
        self._fitted = True
        return self.lml

    def posterior(self, x, want_var: bool = True):
        """Posterior mean and variance of the latent f at ``x`` as two tensors on the query's device.  First call on
        an unfitted model = ONE fused pass (the query rows ride through the factorisation, ``bgp_fit_predict``) -
        the reference builds K lazily and factorises inside the first ``model(x)`` (``battcellgp_full.py:171-173``).
        When the training tensors live on the engine's GPU nothing crosses PCIe but the 2 x M results."""
        xq = torch.as_tensor(x, dtype=torch.float64)
        if xq.ndim == 1:
            xq = xq.reshape(-1, self._train_inputs[0].shape[1])
        if self.n_devices > 1:
            gp = self._shard()
            n_train, m = self._train_inputs[0].shape[0], xq.shape[0]
            # the fused first pass over the ranks' panels (ShardedExactGP.fit_predict) when the variance is wanted - like the
            # single-GPU path below - and the query block is small next to the matrix: every riding row adds a row to every
            # panel, its broadcast and its store (M_pad x N_pad / world doubles per rank).  A large block (add_time_steps:
            # M ~ N, battgp_full.py:86-96) or a mean-only call takes fit + the right-looking passes over the stored factor
            # (ShardedExactGP.predict: blocks of 4096 query rows, so its accumulator stays 4096 x N_pad per rank; the mean
            # needs the same triangular solve as the variance there - z = L^-1 y is what the sharded fit keeps, not alpha).
            if not self._fitted and want_var and m <= max(1024, n_train // 8):
                gp.set_hyp(self.hyp_vector())
                xt, yt = self._train_inputs[0], self._train_targets
                self.lml, mean, var = gp.fit_predict(xt.detach().cpu().numpy(), yt.detach().cpu().numpy(), xq.detach().cpu().numpy(),
                                                     min_var=MIN_VARIANCE)
                self.jitter = gp.jitter
                self._fitted = True
            else:
                self.fit()
                mean, var = gp.predict(xq.detach().cpu().numpy(), min_var=MIN_VARIANCE)
            return torch.as_tensor(mean, device=xq.device), (torch.as_tensor(var, device=xq.device) if want_var else None)
        eng = self.engine()
        xt, yt = self._train_inputs[0], self._train_targets
        m = xq.shape[0]
        fused = not self._fitted and want_var
        if not fused:
            self.fit()
        try:
            if self._train_on_engine_device(eng):
                xq_d = xq.to(xt.device).contiguous()
                out = torch.empty((2, m), dtype=torch.float64, device=xt.device)
                torch.cuda.current_stream(xt.device).synchronize()  # the engine runs on its own streams
                vptr = out[1].data_ptr() if want_var else None
                if fused:
                    eng.set_hyp(self.hyp_vector())
                    self.lml = eng.fit_predict_device(
                        xt.data_ptr(), yt.data_ptr(), xt.shape[0], xt.shape[1], xq_d.data_ptr(), m, out[0].data_ptr(), vptr, MIN_VARIANCE
                    )
                else:
                    eng.predict_device(xq_d.data_ptr(), m, out[0].data_ptr(), vptr, MIN_VARIANCE)
                mean, var = out[0].to(xq.device), (out[1].to(xq.device) if want_var else None)
            else:
                xq_h = xq.detach().cpu().numpy()
                if fused:
                    eng.set_hyp(self.hyp_vector())
                    self.lml, mean, var = eng.fit_predict(xt.detach().cpu().numpy(), yt.detach().cpu().numpy(), xq_h, True, MIN_VARIANCE)
                elif want_var:
                    mean, var = eng.predict(xq_h, want_var=True, min_var=MIN_VARIANCE)
                else:
                    mean, var = eng.predict(xq_h, want_var=False), None
                mean = torch.as_tensor(mean, device=xq.device)
                var = torch.as_tensor(var, device=xq.device) if var is not None else None
        except Exception:
            if fused:
                self._resident = False
            raise
        if fused:
            self._resident = True
            self.jitter = eng.jitter
            self._fitted = True
        return mean, var

    def __call__(self, x) -> Posterior:
        mean, var = self.posterior(x, want_var=True)
        return Posterior(mean, var, mean.device)

    def posterior_mean(self, x) -> np.ndarray:
        """``no_cov`` path: cross fill + GEMV against alpha, no triangular solve of the query block."""
        mean, _ = self.posterior(x, want_var=False)
        return mean.detach().cpu().numpy()

    def neg_mll(self) -> float:
        """``-mll`` with ``mll = lml / N`` - the loss of ``src/gp/training.py:29-30,39-40``."""
        return -self.fit() / self._train_targets.shape[0]

    def neg_mll_and_raw_grad(self):
        """``(loss, d loss / d raw)`` with ``loss = -lml / N``: the LML gradient comes from the GPU(s) - ``bgp_lml_grad``,
        or ``ShardedExactGP.lml_grad`` for an ``n_devices > 1`` model (the same analytic
        ``1/2 tr((alpha alpha^T - Sigma^-1) dSigma/dtheta)`` over the ranks' panels + one all-reduce) - and the
        raw-parameter chain rule (sigmoid / softplus) is applied here: together what ``loss.backward()`` yields in
        ``src/gp/training.py:41``."""
        n = self._train_targets.shape[0]
        lml = self.fit()
        g_hyp = self._shard().lml_grad() if self.n_devices > 1 else self.engine().lml_grad()
        return -lml / n, -(g_hyp * self.dvalue_draw()) / n

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        if self._sharded is not None:
            self._sharded.close()
            self._sharded = None
        self._fitted = False
        self._resident = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hyper-parameter properties (cell_gp.py:64-194) -----------------------------------------------
    @property
    def noise_variance(self):
        return self.likelihood.noise

    @noise_variance.setter
    def noise_variance(self, value):
        self.likelihood.noise = value

    @property
    def noise_variance_constraint(self):
        return self._c_noise

    @noise_variance_constraint.setter
    def noise_variance_constraint(self, value):
        self._c_noise = constraint_from_range(value)
        self._raw = None

    @property
    def outputscale_wiener(self):
        return torch.tensor(self._outputscale_wiener, dtype=torch.float64)

    @outputscale_wiener.setter
    def outputscale_wiener(self, value):
        self._outputscale_wiener = float(_as_float_array(value)[0])
        self._invalidate()

    @property
    def outputscale_wiener_constraint(self):
        return self._c_sw

    @outputscale_wiener_constraint.setter
    def outputscale_wiener_constraint(self, value):
        self._c_sw = constraint_from_range(value)
        self._raw = None

    @property
    def outputscale_rbf(self):
        return torch.tensor(self._outputscale_rbf, dtype=torch.float64)

    @outputscale_rbf.setter
    def outputscale_rbf(self, value):
        self._outputscale_rbf = float(_as_float_array(value)[0])
        self._invalidate()

    @property
    def outputscale_rbf_constraint(self):
        return self._c_sr

    @outputscale_rbf_constraint.setter
    def outputscale_rbf_constraint(self, value):
        self._c_sr = constraint_from_range(value)
        self._raw = None

    @property
    def lengthscale_rbf(self):
        return torch.tensor(self._lengthscale_rbf, dtype=torch.float64).reshape(1, -1)

    @lengthscale_rbf.setter
    def lengthscale_rbf(self, value):
        v = _as_float_array(value)
        d1 = self.train_inputs[0].shape[1] - 1
        if v.size == 1:
            v = np.full(d1, v[0])
        if v.size != d1:
            raise ValueError(f"lengthscale_rbf needs {d1} values, got {v.size}")
        self._lengthscale_rbf = v.astype(np.float64)
        self._invalidate()

    @property
    def lengthscale_rbf_constraint(self):
        return self._c_ls

    @lengthscale_rbf_constraint.setter
    def lengthscale_rbf_constraint(self, value):
        # NOTE: in the reference this setter writes to the ScaleKernel instead of the RBF kernel
        # when n_devices == 1 (cell_gp.py:194 vs the getter :182), so the lengthscale silently
        # keeps GPyTorch's default Positive() (softplus) constraint.  We reproduce the EFFECTIVE
        # behaviour: the range is recorded but the transform stays softplus.
        self._c_ls_requested = constraint_from_range(value)
