#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run ONLY in the build container:

    python tests/golden/make_golden.py

It imports the one piece of the reference that is importable here (pure numpy):
``src.gp.wiener_kernel_temporal.WienerTemporalKernel`` from ``/root/reference``,
and uses its ``(A, Q)`` to drive the Kalman side of the reference's own test
``tests/gp/test_spatiotemporal_gp.py:218-282`` (``test_compare_stgp_egp``).  The
expected values stored in ``stgp_egp.npz`` are therefore reference-derived: the
oracle's exact GP must match them at the reference's tolerance (1e-6 rel).

Everything else on the hot path needs gpytorch, which is not installed here, so
``oracle_cases.npz`` / ``oracle_n2048.json`` hold *oracle* outputs (regression
vectors for the HIP parity tests; the oracle itself is pinned by the known-answer
tests in ``tests/test_oracle.py`` and by ``stgp_egp.npz``).

``lml_pins.npz`` pins the LOG-MARGINAL LIKELIHOOD independently of the oracle and of the HIP engine:

* ``torch.distributions.MultivariateNormal(0, Sigma).log_prob(y)`` - what the reference evaluates at
  ``src/gp/training.py:27-30,39-40`` through gpytorch's ``MultivariateNormal`` / ``ExactMarginalLogLikelihood``
  (``mll = log_prob / N``) - with ``Sigma`` assembled HERE in torch the way the reference's modules assemble
  it: the integrated-Wiener term following ``src/gp/wiener_kernel.py:10-32`` (``covar_dist`` distance, the
  column loop of ``torch.minimum``), the RBF term by gpytorch's mean-centred quadratic-expansion squared
  distance with ``clamp_min(0)`` and a zeroed self-diagonal, ``div(-2).exp()``, the two ``ScaleKernel``
  factors and ``+ sigma^2 I`` (``src/batt_models/cell_gp.py:27-36``).  Different arithmetic (quadratic
  expansion vs direct differences) and a different LAPACK (torch/MKL vs scipy/OpenBLAS) than the oracle.
* the Kalman filter's summed innovation log-likelihood for the ``stgp_egp`` case, driven by the reference's
  own ``WienerTemporalKernel`` ``(A, Q)`` - no N x N factorisation at all.

Only data (inputs and expected outputs) is written; no reference source travels.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from battgp_amd import synthetic  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402
from oracle.kalman_stgp import KalmanSTGP, wiener_kalman_matrices  # noqa: E402


def _foo(t, x):  # target function of the reference test (test_spatiotemporal_gp.py:14-15)
    return np.cos(0.5 * np.pi * x).sum(axis=1) - np.cos(2 * np.pi * t / 40)


def make_stgp_egp(path: str) -> None:
    sys.path.insert(0, "/root/reference")
    from src.gp.wiener_kernel_temporal import WienerTemporalKernel  # reference, pure numpy

    rng = np.random.default_rng(20250205)
    s_w, s_r, ell, noise = 10.0, 3.0, 2.0, 0.1  # test_spatiotemporal_gp.py:226-230
    dims = 3
    tt = np.unique(rng.uniform(0.0, 10.0, 10))
    s_base = rng.uniform(-5.0, 5.0, (20, dims))
    idx = rng.choice(s_base.shape[0], len(tt))
    st = s_base[idx, :]
    yt = _foo(tt, st)
    xt = np.hstack((tt.reshape(-1, 1), st))
    sq = rng.uniform(-5.0, 5.0, (50, dims))

    ref_kernel = WienerTemporalKernel(outputscale=s_w)
    # the restated (A,Q) must agree with the reference's before we rely on either
    for ts in (0.0, 0.37, 2.5):
        a_ref, q_ref = ref_kernel.get_kalman_matrices(ts)
        a_me, q_me = wiener_kalman_matrices(s_w, ts)
        assert np.array_equal(np.asarray(a_ref, dtype=float), a_me)
        if ts == 0.0:  # the reference returns a length-2 zero VECTOR here (temporal_kernel.py:25)
            assert not np.any(q_ref) and not np.any(q_me)
        else:
            assert np.allclose(q_ref, q_me, rtol=1e-15, atol=0)

    rbf_hyp = np.array([0.0, s_r, ell, ell, ell])
    kf = KalmanSTGP(s_base, rbf_hyp, noise, ref_kernel.get_kalman_matrices)
    means, varis, lmls = [], [], []
    ll = 0.0
    for i in range(len(tt)):
        kf.time_step(tt[i] - kf.t)
        ll += kf.update(st[[i], :], yt[[i]])
        lmls.append(ll)  # joint log p(y_1..y_{i+1}) = LML of the exact GP on the first i+1 points
        m, v = kf.predict(sq)
        means.append(m)
        varis.append(v)
    np.savez(
        path,
        tt=tt,
        s_base=s_base,
        xt=xt,
        yt=yt,
        sq=sq,
        hyp=np.array([noise, s_w, s_r, ell, ell, ell]),
        kalman_mean=np.array(means),
        kalman_var=np.array(varis),
        kalman_lml=np.array(lmls),
    )
    # sanity: the oracle agrees at the reference's tolerance
    hyp = np.array([noise, s_w, s_r, ell, ell, ell])
    for i in range(len(tt)):
        gp = OracleGP(K.KERNEL_BATTGP, hyp, xt[: i + 1], yt[: i + 1]).fit()
        xq = np.hstack((np.full((sq.shape[0], 1), tt[i]), sq))
        m, v = gp.predict(xq)
        assert np.linalg.norm(m - means[i]) < 1e-6 * np.linalg.norm(m)
        assert np.linalg.norm(v - varis[i]) < 1e-6 * np.linalg.norm(v)
        assert abs(gp.lml - lmls[i]) < 1e-6 * abs(lmls[i]), (i, gp.lml, lmls[i])
    print("stgp_egp.npz ok")


# ---- the same equivalence at production size and with the production hyper-parameters ---------------
LONG_CHECKPOINTS = (1, 2, 3, 5, 10, 20, 50, 100, 200, 300, 400, 500, 1000, 1500, 2000)


def make_stgp_long(path: str) -> None:
    """``test_compare_stgp_egp`` (``tests/gp/test_spatiotemporal_gp.py:218-282``) carried from 10 to 500 / 2000 time steps
    on the same kind of 20-point spatial basis, (a) with the test's hyper-parameters (``:226-230``) and (b) with the
    numbers the product ships (``src/config.py:39-43``) on inputs in the production ranges (``src/config.py:99-111``).
    The Kalman side is driven by the reference's own ``WienerTemporalKernel`` ``(A, Q)``
    (``src/gp/wiener_kernel_temporal.py:28-35``) and never forms an N x N matrix: its posterior mean / variance at 50
    query points after ``k`` observations and its summed innovation log-likelihood are what the exact GP on the first
    ``k`` points must reproduce (mean, variance and LML at 1e-6 relative, the reference test's tolerance)."""
    sys.path.insert(0, "/root/reference")
    from src.gp.wiener_kernel_temporal import WienerTemporalKernel  # reference, pure numpy

    rng = np.random.default_rng(20260930)
    out = {"checkpoints": np.array(LONG_CHECKPOINTS, dtype=np.int64)}

    def basis_prod(m):
        return np.column_stack((rng.uniform(-80.0, -5.0, m), rng.uniform(40.0, 95.0, m), rng.uniform(10.0, 45.0, m)))

    cases = []
    # (a) the reference test's hyper-parameters; its target function over a full period of the temporal cosine
    tt = np.unique(rng.uniform(0.0, 40.0, 500))
    s_base = rng.uniform(-5.0, 5.0, (20, 3))
    idx = rng.choice(20, len(tt))
    cases.append(("test500", np.array([0.1, 10.0, 3.0, 2.0, 2.0, 2.0]), tt, s_base, idx, _foo(tt, s_base[idx]),
                  rng.uniform(-5.0, 5.0, (50, 3))))
    # (b) production hyper-parameters, production-shaped inputs (synthetic.make_cell_data's resistance model)
    for name, n in (("prod500", 500), ("prod2000", 2000)):
        tt = np.unique(rng.uniform(0.0, 1200.0, n))
        tt[0] = 0.0
        s_base = basis_prod(20)
        idx = rng.choice(20, len(tt))
        st = s_base[idx]
        yt = 0.012 + 0.002 * np.exp(-(st[:, 2] - 10.0) / 20.0) + 1e-6 * tt + rng.normal(0.0, np.sqrt(synthetic.NOISE_VARIANCE), len(tt))
        cases.append((name, synthetic.HYP_BATTGP.copy(), tt, s_base, idx, yt, basis_prod(50)))

    for name, hyp, tt, s_base, idx, yt, sq in cases:
        noise, s_w, s_r = float(hyp[0]), float(hyp[1]), float(hyp[2])
        ref_kernel = WienerTemporalKernel(outputscale=s_w)
        kf = KalmanSTGP(s_base, np.array([0.0, s_r, *hyp[3:]]), noise, ref_kernel.get_kalman_matrices)
        st = s_base[idx]
        xt = np.hstack((tt.reshape(-1, 1), st))
        lmls, means, varis, steps = [], [], [], []
        ll = 0.0
        worst = [0.0, 0.0, 0.0]
        for i in range(len(tt)):
            kf.time_step(tt[i] - kf.t)
            ll += kf.update(st[[i], :], yt[[i]])
            lmls.append(ll)
            if i + 1 in LONG_CHECKPOINTS:
                m, v = kf.predict(sq)
                steps.append(i + 1)
                means.append(m)
                varis.append(v)
                # the oracle agrees at the reference's tolerance before anything is written
                gp = OracleGP(K.KERNEL_BATTGP, hyp, xt[: i + 1], yt[: i + 1]).fit()
                xq = np.hstack((np.full((sq.shape[0], 1), tt[i]), sq))
                mo, vo = gp.predict(xq, clamp=False)
                assert gp.jitter == 0.0
                e = (abs(gp.lml - ll) / abs(ll), np.linalg.norm(mo - m) / np.linalg.norm(m), np.linalg.norm(vo - v) / np.linalg.norm(v))
                assert max(e) < 1e-6, (name, i + 1, e)
                worst = [max(a, b) for a, b in zip(worst, e)]
        print(f"  {name}: {len(tt)} steps, oracle vs Kalman worst rel: lml {worst[0]:.1e} mean {worst[1]:.1e} var {worst[2]:.1e}")
        out[name + "_hyp"] = hyp
        out[name + "_xt"] = xt
        out[name + "_yt"] = yt
        out[name + "_sq"] = sq
        out[name + "_steps"] = np.array(steps, dtype=np.int64)
        out[name + "_kalman_lml"] = np.array(lmls)
        out[name + "_kalman_mean"] = np.array(means)
        out[name + "_kalman_var"] = np.array(varis)
    np.savez_compressed(path, **out)
    print("stgp_egp_long.npz ok")


# ---- LML pins: torch.distributions on a covariance assembled the gpytorch way ------------------------
def _gpytorch_sq_dist(x1, x2, x1_eq_x2):
    """gpytorch ``Kernel.covar_dist(square_dist=True)``: mean-centred quadratic expansion, self-diagonal
    zeroed, clamped at 0 (restated from the gpytorch >= 1.11 sources the reference depends on)."""
    import torch

    adj = x1.mean(-2, keepdim=True)
    x1 = x1 - adj
    x2 = x2 - adj
    n1 = x1.pow(2).sum(-1, keepdim=True)
    n2 = x2.pow(2).sum(-1, keepdim=True)
    a = torch.cat([-2.0 * x1, n1, torch.ones_like(n1)], dim=-1)
    b = torch.cat([x2, torch.ones_like(n2), n2], dim=-1)
    res = a.matmul(b.transpose(-2, -1))
    if x1_eq_x2:
        res.diagonal(dim1=-2, dim2=-1).fill_(0)
    return res.clamp_min_(0)


def _torch_kernel(kernel_id, hyp, x1, x2=None):
    """Noise-free K(x1, x2) in torch, assembled like the reference's kernel modules (x2=None: x1 with itself)."""
    import torch

    same = x2 is None
    x1 = torch.as_tensor(x1, dtype=torch.float64)
    x2 = x1 if same else torch.as_tensor(x2, dtype=torch.float64)
    if kernel_id == K.KERNEL_BATTGP:
        t1, t2 = x1[:, :1], x2[:, :1]
        # WienerKernel.forward (src/gp/wiener_kernel.py:10-32): distance = covar_dist, minval by a column loop
        distance = _gpytorch_sq_dist(t1, t2, same).clamp_min_(1e-30).sqrt_()
        minval = distance.clone()
        for c in range(x2.shape[0]):
            minval[:, c] = torch.minimum(t1, t2[c]).reshape((-1,))
        wiener = torch.pow(minval, 3) / 3 + distance * torch.pow(minval, 2) / 2
        ls = torch.as_tensor(hyp[3:], dtype=torch.float64).reshape(1, -1)
        rbf = _gpytorch_sq_dist(x1[:, 1:].div(ls), x2[:, 1:].div(ls), same).div(-2).exp()
        return hyp[1] * wiener + hyp[2] * rbf
    if kernel_id == K.KERNEL_SCALED_RBF:
        return hyp[1] * _gpytorch_sq_dist(x1.div(float(hyp[2])), x2.div(float(hyp[2])), same).div(-2).exp()
    raise ValueError(kernel_id)


def _torch_sigma(kernel_id, hyp, x):
    import torch

    return _torch_kernel(kernel_id, hyp, x) + float(hyp[0]) * torch.eye(len(x), dtype=torch.float64)


def make_lml_pins(path: str) -> None:
    import torch

    out = {}
    for n in (10, 64, 512):
        x, y = synthetic.make_cell_data(n, seed=7000 + n)
        xs = synthetic.standardise(x)
        cases = {
            "k0prod": (K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y),
            # the hyper-parameters of the reference's own exact-GP test (test_spatiotemporal_gp.py:226-230)
            "k0test": (K.KERNEL_BATTGP, np.array([0.1, 10.0, 3.0, 2.0, 2.0, 2.0]), np.column_stack((x[:, 0] / 120.0, xs[:, 1:])), (y - y.mean()) * 1e3),
            "k1": (K.KERNEL_SCALED_RBF, np.array([3.0, 3.0, 2.0]), xs, (y - y.mean()) * 1e3),
        }
        for name, (kid, hyp, xx, yy) in cases.items():
            sigma = _torch_sigma(kid, hyp, xx)
            mvn = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), covariance_matrix=sigma)
            lml = float(mvn.log_prob(torch.as_tensor(yy, dtype=torch.float64)))
            # posterior of the latent f by an LU solve (torch.linalg.solve - no Cholesky anywhere):
            # mean = K_*X Sigma^-1 y, var = diag(K_**) - diag(K_*X Sigma^-1 K_X*)   (battcellgp_full.py:171-180)
            xq = xx[:: max(1, n // 16)] * 0.999 + 0.001 * xx.mean(axis=0)
            kxs = _torch_kernel(kid, hyp, xx, xq)
            sol = torch.linalg.solve(sigma, torch.cat([torch.as_tensor(yy, dtype=torch.float64).reshape(-1, 1), kxs], dim=1))
            mean_t = (kxs.T @ sol[:, 0]).numpy()
            # prior variance at the queries: the diag=True branch of the kernels (wiener_kernel.py:15-16)
            kss = torch.as_tensor(K.kernel_diag(kid, hyp, xq))
            var_t = (kss - (kxs * sol[:, 1:]).sum(dim=0)).numpy()
            gp = OracleGP(kid, hyp, xx, yy).fit()
            m_o, v_o = gp.predict(xq, clamp=False)
            assert gp.jitter == 0.0
            assert abs(gp.lml - lml) < 1e-6 * abs(lml), (name, n, gp.lml, lml)
            assert np.linalg.norm(m_o - mean_t) < 1e-6 * np.linalg.norm(mean_t), (name, n)
            assert np.max(np.abs(v_o - var_t)) < 1e-6 * np.max(np.abs(var_t)) + 1e-9 * float(hyp[1] if kid else hyp[2]), (name, n)
            print(f"  pin {name} n={n}: lml torch {lml:.12g} oracle {gp.lml:.12g} rel {abs(gp.lml - lml) / abs(lml):.1e}; "
                  f"mean rel {np.linalg.norm(m_o - mean_t) / np.linalg.norm(mean_t):.1e}, var abs {np.max(np.abs(v_o - var_t)):.1e}")
            p = f"{name}_n{n}_"
            out[p + "kernel_id"] = np.int64(kid)
            out[p + "hyp"] = np.asarray(hyp, dtype=np.float64)
            out[p + "x"] = xx
            out[p + "y"] = yy
            out[p + "lml_torch_mvn"] = np.float64(lml)
            out[p + "xq"] = xq
            out[p + "mean_torch_solve"] = mean_t
            out[p + "var_torch_solve"] = var_t
    np.savez_compressed(path, **out)
    print("lml_pins.npz ok")


def make_grad_pins(path: str) -> None:
    """``d lml / d theta`` of the production kernel by torch AUTOGRAD through ``MultivariateNormal.log_prob`` of the
    torch-assembled covariance - the computation behind ``loss.backward()`` at ``src/gp/training.py:39-41`` (there through
    gpytorch's modules, here through the restated assembly above with the hyper-parameters as differentiable tensors).
    Independent of the oracle's and the engine's closed-form trace formula ``1/2 tr((alpha alpha^T - Sigma^-1) dSigma)``."""
    import torch

    out = {}
    for n in (10, 64, 256):
        x, y = synthetic.make_cell_data(n, seed=9000 + n)
        xs = synthetic.standardise(x)
        cases = {
            "k0prod": (K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y),
            "k0test": (K.KERNEL_BATTGP, np.array([0.1, 10.0, 3.0, 2.0, 2.0, 2.0]), np.column_stack((x[:, 0] / 120.0, xs[:, 1:])), (y - y.mean()) * 1e3),
            "k1": (K.KERNEL_SCALED_RBF, np.array([3.0, 3.0, 2.0]), xs, (y - y.mean()) * 1e3),
        }
        for name, (kid, hyp, xx, yy) in cases.items():
            th = torch.tensor(hyp, dtype=torch.float64, requires_grad=True)
            xt = torch.as_tensor(xx, dtype=torch.float64)
            if kid == K.KERNEL_BATTGP:
                t = xt[:, :1]
                distance = _gpytorch_sq_dist(t, t, True).clamp_min(1e-30).sqrt()
                minval = torch.minimum(t, t.T)  # (the column loop of wiener_kernel.py:21-22, vectorised)
                wiener = minval.pow(3) / 3 + distance * minval.pow(2) / 2
                z = xt[:, 1:] / th[3:].reshape(1, -1)
                kmat = th[1] * wiener + th[2] * _gpytorch_sq_dist(z, z, True).div(-2).exp()
            else:
                z = xt / th[2]
                kmat = th[1] * _gpytorch_sq_dist(z, z, True).div(-2).exp()
            sigma = kmat + th[0] * torch.eye(n, dtype=torch.float64)
            mvn = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), covariance_matrix=sigma)
            lml = mvn.log_prob(torch.as_tensor(yy, dtype=torch.float64))
            lml.backward()
            grad = th.grad.numpy().copy()
            from oracle.exact_gp import lml_and_grad

            o_lml, o_grad = lml_and_grad(kid, hyp, xx, yy)
            rel = np.max(np.abs(o_grad - grad) / (np.abs(grad) + 1e-6 * np.abs(grad).max()))
            assert abs(o_lml - lml.item()) < 1e-9 * abs(lml.item()) and rel < 1e-5, (name, n, rel, o_grad, grad)
            print(f"  grad pin {name} n={n}: oracle vs autograd, worst component rel {rel:.1e}")
            p = f"{name}_n{n}_"
            out[p + "kernel_id"] = np.int64(kid)
            out[p + "hyp"] = np.asarray(hyp, dtype=np.float64)
            out[p + "x"] = xx
            out[p + "y"] = yy
            out[p + "lml"] = np.float64(lml.item())
            out[p + "grad"] = grad
    np.savez_compressed(path, **out)
    print("grad_pins.npz ok")


def _case(kernel_id, hyp, x, y, xq):
    gp = OracleGP(kernel_id, hyp, x, y).fit()
    mean, var = gp.predict(xq, clamp=False)
    _, cov = gp.predict(xq[: min(8, len(xq))], full_cov=True)
    kc = K.kernel_matrix(kernel_id, hyp, x[:16])
    return dict(
        kernel_id=np.int64(kernel_id),
        hyp=hyp,
        x=x,
        y=y,
        xq=xq,
        lml=np.float64(gp.lml),
        jitter=np.float64(gp.jitter),
        mean=mean,
        var=var,
        cov8=cov,
        k_corner=kc,
        diag_l=np.diag(gp.L)[:16].copy(),
        alpha=gp.alpha,
    )


def make_oracle_cases(path: str) -> None:
    out = {}
    hyp_rbf_test = np.array([3.0, 3.0, 2.0])  # tests/gp/test_standard_models.py:19-21
    for n in (1, 2, 10, 64, 512):
        x, y = synthetic.make_cell_data(n, seed=1000 + n)
        xq = synthetic.make_query(x, m=37)
        xs = synthetic.standardise(x) if n > 1 else x * 0.0
        xqs = (xq - x.mean(axis=0)) / (x.std(axis=0) if n > 1 else 1.0)
        cases = {
            "k0": _case(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y, xq),
            "k1": _case(K.KERNEL_SCALED_RBF, hyp_rbf_test, xs, y * 1e3, xqs),
            "k2": _case(K.KERNEL_MATERN32, synthetic.HYP_MATERN32, x, y, xq),
            "k3": _case(
                K.KERNEL_ARD_RBF, np.array([2.33e-6, 0.0099, 300.0, 12.11, 33.75, 45.14]), x, y, xq
            ),
        }
        for kname, c in cases.items():
            for field, val in c.items():
                out[f"{kname}_n{n}_{field}"] = val
    np.savez_compressed(path, **out)
    print("oracle_cases.npz ok")


def make_n2048(path: str) -> None:
    n = 2048
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x)
    res = {}
    for name, kid, hyp, xx, xxq in (
        ("k0", K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, xq),
        ("k2", K.KERNEL_MATERN32, synthetic.HYP_MATERN32, x, xq),
    ):
        gp = OracleGP(kid, hyp, xx, y).fit()
        mean, var = gp.predict(xxq, clamp=False)
        res[name] = dict(
            n=n,
            seed=n,
            lml=gp.lml,
            jitter=gp.jitter,
            mean_sum=float(mean.sum()),
            mean_min=float(mean.min()),
            mean_max=float(mean.max()),
            var_sum=float(var.sum()),
            var_min=float(var.min()),
            var_max=float(var.max()),
            mean_first=[float(v) for v in mean[:4]],
            var_first=[float(v) for v in var[:4]],
        )
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("oracle_n2048.json ok")


def make_sklearn_pins(path: str) -> None:
    """An independent third-party exact GP on the kernels the reference cannot pin (it has no Matern at all, and its
    RBF models only through gpytorch, which is absent here): scikit-learn's ``GaussianProcessRegressor`` (Rasmussen &
    Williams alg. 2.1 on LAPACK, no jitter, ``optimizer=None``) with ``ConstantKernel * Matern(nu=1.5) / RBF``:

    * ``log_marginal_likelihood(theta, eval_gradient=True)`` with a ``WhiteKernel`` noise term: the LML and its gradient
      with respect to ``log`` of every hyper-parameter (stored converted to ``d lml / d theta`` in the C-ABI's layout
      ``[noise, s, l_1 ..]``);
    * ``predict(return_std=True)`` with the noise as ``alpha`` (training diagonal only), i.e. the posterior of the LATENT
      function like ``battcellgp_full.py:171-180``.

    Nothing of this repository's oracle or engine takes part in the stored numbers; the oracle is checked against them
    before anything is written."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

    from oracle.exact_gp import lml_and_grad

    out = {}
    worst = {"lml": 0.0, "mean": 0.0, "var": 0.0, "grad": 0.0}
    cases = [
        ("k2", K.KERNEL_MATERN32, synthetic.HYP_MATERN32, False),
        ("k2b", K.KERNEL_MATERN32, np.array([1e-4, 0.5, 300.0, 20.0, 15.0, 12.0]), False),
        ("k3", K.KERNEL_ARD_RBF, np.array([2.33e-6, 0.0099, 500.0, 12.11, 33.75, 45.14]), False),
        ("k1", K.KERNEL_SCALED_RBF, np.array([2.33e-6, 0.0099, 3.0]), True),
    ]
    for name, kid, hyp, standardised in cases:
        for n in (10, 64, 400):
            x, y = synthetic.make_cell_data(n, seed=500 + n)
            xq = synthetic.make_query(x, 24)
            if standardised:
                mu, sd = x.mean(axis=0), x.std(axis=0)
                x, xq = (x - mu) / sd, (xq - mu) / sd
            noise_var, s = hyp[0], hyp[1]
            ls = np.full(x.shape[1], hyp[2]) if kid == K.KERNEL_SCALED_RBF else hyp[2:]
            if kid == K.KERNEL_MATERN32:
                base = Matern(length_scale=ls, nu=1.5)
            elif kid == K.KERNEL_SCALED_RBF:
                base = RBF(length_scale=float(hyp[2]))
            else:
                base = RBF(length_scale=ls)
            # (1) LML + gradient: noise inside the kernel
            kern = ConstantKernel(s) * base + WhiteKernel(noise_var)
            gpr = GaussianProcessRegressor(kernel=kern, alpha=0.0, optimizer=None, normalize_y=False).fit(x, y)
            lml, g_log = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
            # theta = log of [constant_value, length_scale(s), noise_level]
            n_ls = 1 if kid == K.KERNEL_SCALED_RBF else x.shape[1]
            grad = np.empty(hyp.size)
            grad[0] = g_log[-1] / noise_var
            grad[1] = g_log[0] / s
            grad[2:] = g_log[1 : 1 + n_ls] / hyp[2:]
            # (2) latent posterior: noise on the training diagonal only
            gpl = GaussianProcessRegressor(kernel=ConstantKernel(s) * base, alpha=noise_var, optimizer=None, normalize_y=False).fit(x, y)
            mean, std = gpl.predict(xq, return_std=True)
            var = std * std
            # the oracle agrees before anything is written
            gp = OracleGP(kid, hyp, x, y).fit()
            o_mean, o_var = gp.predict(xq, clamp=False)
            _, o_grad = lml_and_grad(kid, hyp, x, y)
            assert gp.jitter == 0.0
            e = {
                "lml": abs(gp.lml - lml) / abs(lml),
                "mean": np.abs(o_mean - mean).max() / np.abs(mean).max(),
                "var": np.abs(o_var - var).max() / s,
                "grad": np.max(np.abs(o_grad - grad) / (np.abs(grad) + 1e-6 * np.abs(grad).max())),
            }
            assert e["lml"] < 1e-9 and e["mean"] < 1e-7 and e["var"] < 1e-7 and e["grad"] < 1e-5, (name, n, e)
            for k_, v in e.items():
                worst[k_] = max(worst[k_], float(v))
            pre = f"{name}_n{n}_"
            out[pre + "kernel_id"] = np.int64(kid)
            out[pre + "hyp"] = hyp
            out[pre + "x"] = x
            out[pre + "y"] = y
            out[pre + "xq"] = xq
            out[pre + "lml"] = np.float64(lml)
            out[pre + "grad"] = grad
            out[pre + "mean"] = mean
            out[pre + "var"] = var
    np.savez_compressed(path, **out)
    print("sklearn_pins.npz ok; oracle vs scikit-learn, worst:", {k_: f"{v:.1e}" for k_, v in worst.items()})


def make_system_contract(path: str) -> None:
    """What the reference's importable pure-pandas objects around the hot path answer on fixed inputs
    (``src/batt_models/battgp.py:16-128``: ``BattGPResult.get_cell_data``, the operating point a ``BattGP`` picks per
    ``RefStrategy``; ``src/operating_point.py``; ``src/batt_models/cellnr.py``) - stored as data for
    ``tests/test_reference_consumers.py::test_system_contract_golden``."""
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, os.path.dirname(HERE))
    import contextlib
    import io

    import pandas as pd  # noqa: F401
    from src.batt_models import battgp as ref_battgp
    from src.batt_models import cellnr as ref_cellnr
    from src.batt_models.ref_strategy import RefStrategy as RefRefStrategy
    from src.operating_point import Op as RefOp
    from test_reference_consumers import CELL_DATA_CALLS, _Data, contract_frame  # the inputs (this repository's)

    frame_args = dict(n_rows=7, cells=[1, 2, 3, 4], seed=3)
    df = contract_frame(**frame_args)
    result = ref_battgp.BattGPResult(None, [], RefOp(-15.0, 90.0, 25.0), df)
    cell_data = []
    for cellnrs, signals, causal, missing in CELL_DATA_CALLS:
        try:
            out = result.get_cell_data(cellnrs, signals, causal, missing)
        except ValueError:
            cell_data.append("ValueError")
            continue
        # which source column each output column carries: identified by VALUE (every column of the frame is distinct)
        sources = []
        for pos in range(out.shape[1]):
            hits = [c for c in df.columns if np.array_equal(df[c].to_numpy(), out.iloc[:, pos].to_numpy())]
            assert len(hits) == 1
            sources.append(hits[0])
        cell_data.append({"columns": list(out.columns), "sources": sources})
    ops = [(-15.0, 90.0, 25.0), (-27.123456, 73.5, 18.004), (0, 100, -5)]
    bd = _Data("g", n_cells=1)
    picks = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for strategy in ("mean", "median"):
            picks[strategy] = ref_battgp.BattGP(bd, ref_strategy=RefRefStrategy(strategy)).ref_op.disp_str()
        manual = ref_battgp.BattGP(bd, ref_strategy=RefRefStrategy(RefOp(1.0, 2.0, 3.0))).ref_op.disp_str()
    gold = {
        "frame": frame_args,
        "frame_columns": list(df.columns),
        "cell_data": cell_data,
        "op": [[list(v), RefOp(*v).disp_str(), repr(RefOp(*v))] for v in ops],
        "cell_tags": [[c, ref_cellnr.get_cell_tag(c)] for c in (-1, 0, 1, 7, 12, 108)],
        "causal_tags": [ref_cellnr.get_causal_tag(True), ref_cellnr.get_causal_tag(False)],
        "strategy_picks": picks,
        "manual_pick": manual,
    }
    with open(path, "w") as f:
        json.dump(gold, f, indent=1)
    print("system_contract.json ok")


if __name__ == "__main__":
    make_system_contract(os.path.join(HERE, "system_contract.json"))
    make_sklearn_pins(os.path.join(HERE, "sklearn_pins.npz"))
    make_grad_pins(os.path.join(HERE, "grad_pins.npz"))
    make_stgp_egp(os.path.join(HERE, "stgp_egp.npz"))
    make_stgp_long(os.path.join(HERE, "stgp_egp_long.npz"))
    make_lml_pins(os.path.join(HERE, "lml_pins.npz"))
    make_oracle_cases(os.path.join(HERE, "oracle_cases.npz"))
    make_n2048(os.path.join(HERE, "oracle_n2048.json"))
