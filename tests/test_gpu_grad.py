"""GPU parity, part 4: the LML GRADIENT (bgp_lml_grad: Sigma^-1 formed in place over the factor + the fused reduction
pass; the computation behind ``loss.backward()`` at /root/reference/src/gp/training.py:39-41).

Every `-m gpu` test that calls ``lml_grad`` on the single-GPU engine lives HERE, and tests/conftest.py::GPU_ORDER
places this module after the fit / predict / natural-size / slab-layout modules: the driver runs the suite with
``-x``, and a fault in the (younger) gradient kernels must not leave BASELINE configs 2 and 3 unreached.
tests/test_cabi_and_host.py::test_gradient_tests_sit_behind_the_core_gpu_modules checks that ordering statically; the
dynamic rehearsal is `pytest -m gpu -x --emu --emu-fault bgp_lml_grad`."""

import os

import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from problem_gen import LA_MEASURED, check_problem, problems  # noqa: E402

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402

REL = 1e-6  # north_star tolerance (LML, posterior mean)


# ---------------------------------------------------------------------------------------------
# against the oracle's analytic gradient
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_SCALED_RBF, K.KERNEL_MATERN32, K.KERNEL_ARD_RBF])
@pytest.mark.parametrize("n", [24, 700])
def test_lml_gradient_matches_oracle(kid, n):
    from oracle.exact_gp import lml_and_grad

    rng = np.random.default_rng(n + kid)
    x = np.column_stack([np.sort(rng.uniform(0, 5, n)), rng.normal(size=(n, 3))])
    y = rng.normal(size=n)
    hyp = {
        K.KERNEL_BATTGP: np.array([0.1, 0.5, 1.3, 0.8, 1.1, 1.7]),
        K.KERNEL_SCALED_RBF: np.array([0.1, 1.3, 1.5]),
        K.KERNEL_MATERN32: np.array([0.1, 1.3, 2.0, 0.8, 1.1, 1.7]),
        K.KERNEL_ARD_RBF: np.array([0.1, 1.3, 2.0, 0.8, 1.1, 1.7]),
    }[kid]
    lml_ref, g_ref = lml_and_grad(kid, hyp, x, y)
    e = ExactGPEngine(kid, hyp)
    lml = e.fit(x, y)
    m0, v0 = e.predict(x[:5] + 0.01, min_var=-1.0)
    d0 = e.factor_diag()
    bytes0 = e.device_bytes()
    g = e.lml_grad()
    bytes1 = e.device_bytes()
    g2 = e.lml_grad()  # workspaces are reused; result is run-to-run identical
    # the gradient forms Sigma^-1 IN PLACE over the factor; the next call that needs L gets it back bit for bit
    m, v = e.predict(x[:5] + 0.01, min_var=-1.0)
    d1 = e.factor_diag()
    alpha = e.alpha()
    e.close()
    assert abs(lml - lml_ref) < 1e-9 * abs(lml_ref)
    assert np.allclose(g, g_ref, rtol=1e-7, atol=1e-9 * np.abs(g_ref).max()), (g, g_ref)
    assert np.array_equal(g, g2)
    assert np.array_equal(m, m0) and np.array_equal(v, v0) and np.array_equal(d0, d1)
    ref = OracleGP(kid, hyp, x, y).fit()
    assert np.linalg.norm(alpha - ref.alpha) < 1e-8 * np.linalg.norm(ref.alpha)
    # no second N^2 buffer: only panel-sized workspaces may have been added (two transposed row blocks + the panel inverses)
    npad = -(-n // 64) * 64
    assert bytes1 - bytes0 <= 8 * (4 * (npad + 64) * 512 + 4 * 512 * 512) + 4096, (bytes0, bytes1)


def test_lml_gradient_production_hyperparameters():
    from oracle.exact_gp import lml_and_grad

    x, y = synthetic.make_cell_data(1500, seed=4)
    lml_ref, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.fit(x, y)
    g = e.lml_grad()
    e.close()
    # entries span 20 orders of magnitude (d/ds_w ~ 1e13): compare each relative to itself
    assert np.allclose(g, g_ref, rtol=1e-5), (g, g_ref)

@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(problems(n_max=3000, m_max=400, noise_lo=-3.0, la_words=LA_MEASURED))
def test_random_problems_gradient_matches_the_oracle(prob):
    """the property of test_gpu_parity.py::test_random_problems_match_the_oracle (same derandomised problems: ragged N,
    D = 1..6, all kernels, duplicated points, both panel schemes, slab layout) with the analytic gradient added"""
    check_problem(*prob, grad=True)


# ---------------------------------------------------------------------------------------------
# against third-party pins (scikit-learn, torch autograd)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["k2", "k2b", "k3", "k1"])
@pytest.mark.parametrize("n", [10, 64, 400])
def test_gradient_matches_scikit_learn_pins(golden_dir, name, n):
    """The gradient half of tests/golden/sklearn_pins.npz (make_golden.py::make_sklearn_pins; the LML / posterior half
    is asserted in test_gpu_pins_and_sizes.py): scikit-learn's analytic ``log_marginal_likelihood(eval_gradient=True)``
    for Matern-3/2 and the RBF kernels; and the posterior after the gradient equals the one before it bit for bit
    (the factor the gradient consumed is restored on demand)."""
    g = np.load(os.path.join(golden_dir, "sklearn_pins.npz"))
    p = f"{name}_n{n}_"
    kid, hyp, x, y, xq = int(g[p + "kernel_id"]), g[p + "hyp"], g[p + "x"], g[p + "y"], g[p + "xq"]
    wgrad = g[p + "grad"]
    e = ExactGPEngine(kid, hyp)
    try:
        e.fit(x, y)
        m, v = e.predict(xq, min_var=-1.0)
        grad = e.lml_grad()
        m_after, v_after = e.predict(xq, min_var=-1.0)
    finally:
        e.close()
    assert np.all(np.abs(grad - wgrad) <= 1e-5 * np.abs(wgrad) + 1e-7 * np.abs(wgrad).max()), (grad, wgrad)
    assert np.array_equal(m, m_after) and np.array_equal(v, v_after)


@pytest.mark.parametrize("name", ["k0prod", "k0test", "k1"])
@pytest.mark.parametrize("n", [10, 64, 256])
def test_lml_gradient_matches_torch_autograd_pins(golden_dir, name, n):
    """bgp_lml_grad (Sigma^-1 in place over the factor + the fused reduction pass) against torch AUTOGRAD through
    MultivariateNormal.log_prob of the torch-assembled covariance - the computation behind loss.backward() at
    src/gp/training.py:39-41 - for the production kernel with the production and the reference test's hyper-parameters
    and for ScaledRBFModel's kernel (make_golden.py::make_grad_pins); full square and column slabs."""
    g = np.load(os.path.join(golden_dir, "grad_pins.npz"))
    p = f"{name}_n{n}_"
    kid, hyp, x, y, want = int(g[p + "kernel_id"]), g[p + "hyp"], g[p + "x"], g[p + "y"], g[p + "grad"]
    for slab in (-1, 64):
        e = ExactGPEngine(kid, hyp)
        try:
            if slab > 0:
                e.set_options(nb_outer=slab)  # (a slab is a whole number of outer panels)
            e.set_layout(slab)
            lml = e.fit(x, y)
            grad = e.lml_grad()
        finally:
            e.close()
        assert abs(lml - g[p + "lml"]) <= REL * abs(g[p + "lml"])
        assert np.all(np.abs(grad - want) <= 1e-5 * np.abs(want) + 1e-7 * np.abs(want).max()), (slab, grad, want)


def test_slab_layout_gradient():
    from oracle.exact_gp import lml_and_grad

    n = 1500
    x, y = synthetic.make_cell_data(n, seed=11)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.set_layout(512)
    lml = e.fit(x, y)
    g = e.lml_grad()
    e.close()
    lml_ref, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    assert abs(lml - lml_ref) <= REL * abs(lml_ref)
    assert np.allclose(g, g_ref, rtol=1e-5), (g, g_ref)


# ---------------------------------------------------------------------------------------------
# natural size (BASELINE config 2)
# ---------------------------------------------------------------------------------------------
def _gradient_vs_lml_differences(e, hyp, rel_step=2e-4, rtol=2e-3):
    """d lml / d log(theta_i) from bgp_lml_grad against central differences of the engine's own (oracle-checked) LML along
    each log-parameter: 2 resident re-fits per parameter.  Not an independent pin (those are the scikit-learn / autograd
    pins at small N) - the check that the in-place inverse and the reduction pass stay consistent with the value at a size
    where the automatic defaults switch code paths."""
    grad = e.lml_grad()
    assert np.all(np.isfinite(grad))
    for i in range(hyp.size):
        hp, hm = hyp.copy(), hyp.copy()
        hp[i] *= 1.0 + rel_step
        hm[i] *= 1.0 - rel_step
        fd = (e.refit(hp) - e.refit(hm)) / (2.0 * rel_step)  # d lml / d log theta_i
        an = grad[i] * hyp[i]
        # (central differences of a value that is itself good to ~1e-10 relative: an absolute floor from that noise)
        assert abs(fd - an) <= rtol * abs(an) + 2e-9 * abs(e.lml) / rel_step + 1e-3, (i, fd, an)
    e.refit(hyp)
    return grad


@pytest.mark.gpu_sized
def test_n40000_gradient_at_the_natural_size():
    """BASELINE config 2's size: the in-place-inverse gradient under the automatic defaults (panel scheme 1, NB = 1024,
    full square) and in column slabs - consistent with differences of the LML, identical between the two layouts up to
    the summation order of the reduction pass, and the factor comes back bit for bit."""
    n = 40000
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, 300)
    hyp = synthetic.HYP_BATTGP.copy()
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    try:
        lml = e.fit(x, y)
        mean, var = e.predict(xq, min_var=-1.0)
        g_full = _gradient_vs_lml_differences(e, hyp)
        mean2, var2 = e.predict(xq, min_var=-1.0)
        assert e.lml == lml and np.array_equal(mean, mean2) and np.array_equal(var, var2)
        e.set_layout(8192)
        assert e.fit(x, y) == lml and e.layout()[0] == 8192
        g_slab = e.lml_grad()
        assert np.allclose(g_slab, g_full, rtol=1e-9), (g_slab, g_full)
        mean3, var3 = e.predict(xq, min_var=-1.0)
        assert np.array_equal(mean, mean3) and np.array_equal(var, var3)
    finally:
        e.close()


# ---------------------------------------------------------------------------------------------
# what a gradient leaves behind: phase-time bookkeeping, the opt-in kept factor, the reduction's block bounds
# ---------------------------------------------------------------------------------------------
def test_restoring_the_factor_does_not_overwrite_the_fits_phase_times():
    """the re-run of the fit that brings the factor back after a gradient consumed it reports its cost in its
    own slot (restore_ms); fill / potrf / trailing-update figures stay those of the fit the caller asked for, and the
    gradient pass has its own slot too (grad_ms; it used to share solve_ms)."""
    x, y = synthetic.make_cell_data(1500, seed=3)
    xq = synthetic.make_query(x, 20)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    try:
        e.fit(x, y)
        m0, v0 = e.predict(xq, min_var=-1.0)
        t_fit = e.phase_times()
        assert t_fit["grad_ms"] == 0.0 and t_fit["restore_ms"] == 0.0
        e.lml_grad()
        t_grad = e.phase_times()
        assert t_grad["grad_ms"] > 0.0 and t_grad["restore_ms"] == 0.0
        m1, v1 = e.predict(xq, min_var=-1.0)  # needs L: re-runs the fit
        t_after = e.phase_times()
    finally:
        e.close()
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    assert t_after["restore_ms"] > 0.0
    for k in ("fill_ms", "potrf_ms", "trail_ms", "trail_flop", "fill_bytes", "trail_launches", "trail_union_ms", "h2d_ms"):
        assert t_after[k] == t_fit[k], k


def test_a_failed_restoring_fit_leaves_the_fits_phase_times_alone():
    """when the re-run that brings the factor back FAILS (here: the jitter rung the fit needed is no longer
    allowed), the fit's FILL / POTRF / TRAIL* slots still hold what the caller's fit measured, the failed attempt is in
    restore_ms, and the model recovers with the next fit."""
    from battgp_amd.engine import NotPSDError, NumericalWarning

    rng = np.random.default_rng(5)
    n = 300
    x = rng.normal(size=(n, 2))
    x[150:] = x[:150]  # every point twice and no noise: singular without jitter
    y = rng.normal(size=n)
    hyp = np.array([0.0, 1.0, 1.0, 1.0])
    e = ExactGPEngine(K.KERNEL_ARD_RBF, hyp)
    try:
        e.set_options(nb_outer=128)
        with pytest.warns(NumericalWarning):
            e.fit(x, y)
        assert e.jitter > 0.0
        t_fit = e.phase_times()
        e.lml_grad()  # consumes the factor
        t_grad = e.phase_times()
        e.set_options(max_tries=0)  # the plain attempt only: the restoring re-run can not succeed
        with pytest.raises(NotPSDError):
            e.predict(x[:5])
        t_after = e.phase_times()
        for k in ("fill_ms", "potrf_ms", "trail_ms", "trail_flop", "fill_bytes", "trail_launches", "trail_union_ms", "h2d_ms"):
            assert t_after[k] == t_fit[k], k
        # the failed re-run never reaches the solve: that slot is still the gradient's alpha solve (and is not counted as spent
        # by the re-run)
        assert t_after["solve_ms"] == t_grad["solve_ms"]
        assert t_after["restore_ms"] > 0.0
        e.set_options(max_tries=3)
        with pytest.warns(NumericalWarning):
            e.fit(x, y)
        mean, _ = e.predict(x[:5])
        assert np.all(np.isfinite(mean))
    finally:
        e.close()


@pytest.mark.parametrize("slab", [-1, 512])
def test_keep_factor_brings_the_factor_back_by_a_copy(slab):
    """bgp_set_keep_factor: same gradient, same posterior bit for bit, one more factor-sized buffer while it is on."""
    x, y = synthetic.make_cell_data(1500, seed=5)
    xq = synthetic.make_query(x, 20)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    try:
        e.set_layout(slab)
        lml = e.fit(x, y)
        m0, v0 = e.predict(xq, min_var=-1.0)
        d0 = e.factor_diag()
        g_plain = e.lml_grad()
        e.predict(xq)
        b_plain = e.device_bytes()
        e.set_keep_factor(True)
        g_keep = e.lml_grad()
        assert e.device_bytes() - b_plain == e.layout()[1]
        m1, v1 = e.predict(xq, min_var=-1.0)
        assert e.lml == lml
        g_again = e.lml_grad()  # the kept copy is still the current factor: no second save
        d1 = e.factor_diag()
        alpha = e.alpha()
        # a new factor invalidates the copy
        hyp2 = synthetic.HYP_BATTGP * np.array([1.5, 0.7, 1.2, 0.9, 1.1, 0.8])
        e.refit(hyp2)
        g2 = e.lml_grad()
        m2, _ = e.predict(xq, min_var=-1.0)
        e.set_keep_factor(False)
        assert e.device_bytes() == b_plain
    finally:
        e.close()
    assert np.array_equal(g_plain, g_keep) and np.array_equal(g_plain, g_again)
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1) and np.array_equal(d0, d1)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    assert np.linalg.norm(alpha - ref.alpha) < 1e-8 * np.linalg.norm(ref.alpha)
    from oracle.exact_gp import lml_and_grad

    _, g2_ref = lml_and_grad(K.KERNEL_BATTGP, hyp2, x, y)
    assert np.allclose(g2, g2_ref, rtol=1e-5)
    m2_ref, _ = OracleGP(K.KERNEL_BATTGP, hyp2, x, y).fit().predict(xq)
    assert np.linalg.norm(m2 - m2_ref) <= REL * np.linalg.norm(m2_ref)


@pytest.mark.parametrize("nrows,ncols", [(700, 100), (650, 33), (512, 32), (100, 100), (64, 7)])
def test_gradient_reduction_block_reads_nothing_outside_the_described_block(nrows, ncols):
    """bgp_grad_reduce_block_dev walks 512 x 32 tiles; a block that is not a whole number of them must not
    pick up what lies next to it.  The buffer holds NaN outside the described rows / columns, and the sums equal the host
    evaluation of  sum' (alpha_i alpha_j - P_ij) dSigma_ij/d(.)  over the block."""
    import ctypes as C

    from battgp_amd import _lib

    rng = np.random.default_rng(nrows + ncols)
    n, d, r0 = 900, 4, 128
    x = np.column_stack([np.sort(rng.uniform(0, 5, n)), rng.normal(size=(n, 3))])
    alpha = rng.normal(size=n)
    hyp = np.array([0.1, 0.5, 1.3, 0.8, 1.1, 1.7])
    ld = nrows + 6
    P = np.full((ncols + 40, ld), np.nan)  # column-major [ld, ncols + 40]
    blk = rng.normal(size=(nrows, ncols))
    P[:ncols, :nrows] = blk.T
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    lib = _lib.load()
    try:
        nacc = lib.bgp_grad_nacc()
        tx = torch.from_numpy(x).to("cuda")
        tp = torch.from_numpy(P).to("cuda")
        ta = torch.from_numpy(alpha).to("cuda")
        acc = torch.zeros(nacc, dtype=torch.float64, device="cuda")
        vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        torch.cuda.synchronize()  # the uploads and the zero fill ran on torch's stream; the engine has its own
        rc = lib.bgp_grad_reduce_block_dev(e._h, vp(tx), n, d, r0, nrows, ncols, vp(tp), ld, vp(ta), vp(acc), 0)
        assert rc == 0, lib.bgp_last_error(e._h)
        e.sync()
        got = acc.cpu().numpy()
        # refused: a leading dimension shorter than the block
        assert lib.bgp_grad_reduce_block_dev(e._h, vp(tx), n, d, r0, nrows, ncols, vp(tp), nrows - 2, vp(ta), vp(acc), 0) != 0
    finally:
        e.close()
    # host evaluation over the lower part of the block
    ii = r0 + np.arange(nrows)[:, None]
    jj = r0 + np.arange(ncols)[None, :]
    mask = (ii >= jj) & (ii < n) & (jj < n)
    wgt = np.where(ii == jj, 1.0, 2.0)
    W = np.where(mask, wgt * (alpha[np.minimum(ii, n - 1)] * alpha[np.minimum(jj, n - 1)] - blk), 0.0)
    xi, xj = x[np.minimum(ii, n - 1)[:, 0]], x[np.minimum(jj, n - 1)[0]]
    t_i, t_j = xi[:, 0][:, None], xj[:, 0][None, :]
    mn = np.minimum(t_i, t_j)
    wien = mn ** 3 / 3.0 + np.abs(t_i - t_j) * mn ** 2 / 2.0
    u2 = [((xi[:, k][:, None] - xj[:, k][None, :]) / hyp[2 + k]) ** 2 * 0.5 for k in (1, 2, 3)]
    g = np.exp(-(u2[0] + u2[1] + u2[2]))
    want = np.zeros(nacc)
    want[0] = np.sum(np.where(ii == jj, W, 0.0))
    want[1] = np.sum(W * wien)
    want[2] = np.sum(W * g)
    for k in range(3):
        want[4 + k] = np.sum(W * g * u2[k])
    assert np.all(np.isfinite(got)), got
    assert np.allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max()), (got, want)
