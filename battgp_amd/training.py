"""Hyper-parameter optimisation loops of the ``full_gp`` path, same signatures and stopping rules
as ``src/gp/training.py`` (``train_exact_gp_adam`` :11-67, ``train_exact_gp_lbfgs`` :108-171,
``train_exact_gp_botorch`` :70-105).

Each iteration evaluates ``loss = -mll = -lml / N`` with ONE resident re-fit on the GPU
(``bgp_refit``: fill + jittered Cholesky, no re-upload) and its gradient with ``bgp_lml_grad``
(``1/2 tr((alpha alpha^T - Sigma^-1) dSigma/dtheta)``: Sigma^-1 formed IN PLACE over the factor by two more N^3/3
MFMA passes plus one fused reduction pass - no second N^2 buffer, so training reaches the same N as inference; for an
``n_devices > 1`` model the same steps run over the ranks' panels, ``ShardedExactGP.lml_grad``); the chain rule to the raw (unconstrained) parameters - what
``loss.backward()`` gives the reference - is applied on the host (``BatteryCellGP.neg_mll_and_raw_grad``).
"""

from __future__ import annotations

import numpy as np

from .cell_gp import BatteryCellGP


def _loss(model: BatteryCellGP, raw: np.ndarray) -> float:
    model.set_raw_vector(raw)
    return model.neg_mll()


def _loss_and_grad(model: BatteryCellGP, raw: np.ndarray):
    model.set_raw_vector(raw)
    return model.neg_mll_and_raw_grad()


def _loss_and_grad_fd(model: BatteryCellGP, raw: np.ndarray, rel_step: float = 1e-4):
    """Central finite differences of the GPU loss in raw space (2 x nparam re-fits): kept as an
    independent check of the analytic gradient, not used by the trainers."""
    f0 = _loss(model, raw)
    g = np.zeros_like(raw)
    for i in range(raw.size):
        h = rel_step * max(1.0, abs(raw[i]))
        rp, rm = raw.copy(), raw.copy()
        rp[i] += h
        rm[i] -= h
        g[i] = (_loss(model, rp) - _loss(model, rm)) / (2.0 * h)
    model.set_raw_vector(raw)
    return f0, g


def train_exact_gp_adam(
    model: BatteryCellGP,
    train_x=None,
    train_y=None,
    max_iter: int = 100,
    rel_ftol: float = 0.0,
    loss_scale: int = 1,
    lr: float = 1,
    messages: bool = True,
) -> np.ndarray:
    """Adam (torch defaults: betas 0.9/0.999, eps 1e-8) on the raw parameters; the loss history and
    the relative-change stop rule follow ``training.py:35-65`` line by line."""
    model.train()
    model.likelihood.train()
    raw = model.raw_vector()
    m = np.zeros_like(raw)
    v = np.zeros_like(raw)
    b1, b2, eps = 0.9, 0.999, 1e-8
    loss = _loss(model, raw)
    if messages:
        print(f"start loss={loss * loss_scale}")
    losses = np.zeros(max_iter + 1) * np.nan
    for i in range(max_iter):
        loss, grad = _loss_and_grad(model, raw)
        current = loss * loss_scale
        losses[i] = current
        if i > 0:
            prev = losses[i - 1]
            if abs((current - prev) / prev) < rel_ftol:
                losses = losses[: i + 1]
                break
        m = b1 * m + (1 - b1) * grad
        v = b2 * v + (1 - b2) * grad * grad
        mhat = m / (1 - b1 ** (i + 1))
        vhat = v / (1 - b2 ** (i + 1))
        raw = raw - lr * mhat / (np.sqrt(vhat) + eps)
    # the reference re-evaluates the loss of the LAST forward output (before the final step)
    final = loss * loss_scale
    losses[-1] = final
    if messages:
        print(f"final loss={final}")
    model.set_raw_vector(raw)
    model.eval()
    model.likelihood.eval()
    return losses


def train_exact_gp_lbfgs(
    model: BatteryCellGP,
    train_x=None,
    train_y=None,
    max_iter: int = 20,
    rel_ftol: float = 1e-4,
    loss_scale: int = 1,
    lr: float = 1,
    messages: bool = True,
) -> np.ndarray:
    """The reference's ``torch_lbfgs`` trainer (``training.py:108-171``) with the SAME optimiser object:
    ``torch.optim.LBFGS(params, line_search_fn="strong_wolfe", lr=lr)`` - so up to 20 inner quasi-Newton iterations
    per ``step`` (torch's default ``max_iter``), curvature history kept across the outer loop, ``lr`` honoured - driving
    one flat fp64 tensor of raw parameters.  The closure does not back-propagate: loss and gradient of each
    evaluation come from the GPU (``bgp_refit`` + ``bgp_lml_grad``) and are handed to torch as ``.grad``.  Outer loop,
    loss history and the relative-change stop rule as in ``training.py:147-164``."""
    import torch

    model.train()
    model.likelihood.train()
    raw = torch.tensor(model.raw_vector(), dtype=torch.float64, requires_grad=True)
    optimizer = torch.optim.LBFGS([raw], line_search_fn="strong_wolfe", lr=lr)

    def closure():
        optimizer.zero_grad()
        value, grad = _loss_and_grad(model, raw.detach().numpy().copy())
        raw.grad = torch.from_numpy(np.ascontiguousarray(grad, dtype=np.float64))
        return torch.tensor(value, dtype=torch.float64)

    loss = _loss(model, raw.detach().numpy().copy())
    if messages:
        print(f"start loss={loss * loss_scale}")
    losses = np.zeros(max_iter + 1) * np.nan
    for i in range(max_iter):
        loss = _loss(model, raw.detach().numpy().copy())  # cached when the optimiser ended on this point
        current = loss * loss_scale
        losses[i] = current
        if i > 0:
            prev = losses[i - 1]
            if abs((current - prev) / prev) < rel_ftol:
                losses = losses[: i + 1]
                break
        optimizer.step(closure)
    # like the reference: the last recorded value is the loss of the LAST forward pass, before the final step
    losses[-1] = loss * loss_scale
    if messages:
        print(f"final loss={losses[-1]}")
    model.set_raw_vector(raw.detach().numpy().copy())
    model.eval()
    model.likelihood.eval()
    return losses


# options the reference hands to botorch's scipy L-BFGS-B (training.py:82-95); "eps" only matters for
# finite-difference gradients and is unused with an analytic jacobian
BOTORCH_SCIPY_OPTIONS = {"maxiter": 10000, "ftol": 1e-15, "gtol": 1e-15, "maxfun": 10000, "maxls": 10000}


def train_exact_gp_botorch(model: BatteryCellGP, train_x=None, train_y=None, **_kwargs) -> float:
    """``fit_gpytorch_mll`` = scipy L-BFGS-B on the raw parameters, with the reference's options; returns the final
    loss ``-mll`` (``training.py:70-105``, which also prints it under the label "start loss")."""
    from scipy.optimize import minimize

    model.train()
    model.likelihood.train()
    res = minimize(
        lambda r: _loss_and_grad(model, r), model.raw_vector(), jac=True, method="L-BFGS-B", options=dict(BOTORCH_SCIPY_OPTIONS)
    )
    model.set_raw_vector(res.x)
    model.eval()
    model.likelihood.eval()
    loss = model.neg_mll()
    print(f"start loss={loss}")
    return loss
