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
    means, varis = [], []
    for i in range(len(tt)):
        kf.time_step(tt[i] - kf.t)
        kf.update(st[[i], :], yt[[i]])
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
    )
    # sanity: the oracle agrees at the reference's tolerance
    hyp = np.array([noise, s_w, s_r, ell, ell, ell])
    for i in range(len(tt)):
        gp = OracleGP(K.KERNEL_BATTGP, hyp, xt[: i + 1], yt[: i + 1]).fit()
        xq = np.hstack((np.full((sq.shape[0], 1), tt[i]), sq))
        m, v = gp.predict(xq)
        assert np.linalg.norm(m - means[i]) < 1e-6 * np.linalg.norm(m)
        assert np.linalg.norm(v - varis[i]) < 1e-6 * np.linalg.norm(v)
    print("stgp_egp.npz ok")


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


if __name__ == "__main__":
    make_stgp_egp(os.path.join(HERE, "stgp_egp.npz"))
    make_oracle_cases(os.path.join(HERE, "oracle_cases.npz"))
    make_n2048(os.path.join(HERE, "oracle_n2048.json"))
