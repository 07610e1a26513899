"""GPU parity tests: every check drives the HIP path through the C-ABI (libbattgp.so) and
compares with the CPU oracle / committed golden vectors.  Tolerances are the north-star's:
1e-6 relative on LML and posterior mean (fp64); tighter where the algebra allows."""

import json
import os
import sys
import warnings

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from problem_gen import LA_MEASURED, check_problem, problems  # noqa: E402

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine, NotPSDError, NumericalWarning  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402

REL = 1e-6  # north_star tolerance (LML, posterior mean)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def to_dev_colmajor(a: np.ndarray, ld: int | None = None):
    """numpy [m, n] -> torch cuda tensor [n, ld] whose memory is the column-major matrix."""
    m, n = a.shape
    ld = m if ld is None else ld
    t = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    t[:, :m] = torch.from_numpy(np.ascontiguousarray(a.T))
    return t


def from_dev_colmajor(t, m: int) -> np.ndarray:
    return t[:, :m].cpu().numpy().T.copy()


@pytest.fixture(scope="module")
def eng():
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, device=0)
    yield e
    e.close()


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize(
    "m,n,k,lower", [(128, 128, 16, 0), (256, 128, 64, 0), (384, 384, 512, 1), (130, 66, 64, 0), (16, 200, 64, 0), (1024, 640, 128, 1)]
)
def test_gemm_nt_sub(eng, m, n, k, lower):
    rng = np.random.default_rng(m * 7 + n)
    a = rng.normal(size=(m, k))
    b = rng.normal(size=(n, k)) + 0.5  # asymmetric on purpose: catches row/col swaps
    c = rng.normal(size=(m, n))
    lda, ldb, ldc = m + 2, n + 4, m + 6
    ta, tb, tc = to_dev_colmajor(a, lda), to_dev_colmajor(b, ldb), to_dev_colmajor(c, ldc)
    torch.cuda.synchronize()
    eng.gemm_nt_sub_device(tc.data_ptr(), ldc, ta.data_ptr(), lda, tb.data_ptr(), ldb, m, n, k, lower)
    got = from_dev_colmajor(tc, m)
    want = c - a @ b.T
    if lower:
        # tiles (128x128) strictly above the diagonal are not touched
        ti = np.arange(m)[:, None] // 128
        tj = np.arange(n)[None, :] // 128
        mask = ti >= tj
        assert np.allclose(got[mask], want[mask], rtol=1e-12, atol=1e-10)
        assert np.array_equal(got[~mask], c[~mask])
    else:
        assert np.allclose(got, want, rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("n", [64, 128, 320, 1024, 1536])
def test_potrf_dev_random_spd(eng, n):
    rng = np.random.default_rng(n)
    g = rng.normal(size=(n, n))
    a = g @ g.T + n * np.eye(n)
    lda = n + 2
    ta = to_dev_colmajor(a, lda)
    torch.cuda.synchronize()
    info = eng.potrf_device(ta.data_ptr(), n, lda)
    assert info == 0
    l_got = np.tril(from_dev_colmajor(ta, n))
    l_ref = np.linalg.cholesky(a)
    assert rel_err(l_got, l_ref) < 1e-12
    assert rel_err(l_got @ l_got.T, a) < 1e-13


def test_potrf_dev_reports_failing_minor(eng):
    n = 256
    a = np.eye(n)
    a[100, 100] = -1.0
    ta = to_dev_colmajor(a, n)
    torch.cuda.synchronize()
    assert eng.potrf_device(ta.data_ptr(), n, n) == 101


@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_SCALED_RBF, K.KERNEL_MATERN32, K.KERNEL_ARD_RBF])
@pytest.mark.parametrize("n1,n2", [(1, 1), (37, 201), (300, 129)])
def test_kernel_matrix_matches_oracle(kid, n1, n2):
    x1, _ = synthetic.make_cell_data(n1, seed=5)
    x2, _ = synthetic.make_cell_data(n2, seed=6)
    hyp = {
        K.KERNEL_BATTGP: synthetic.HYP_BATTGP,
        K.KERNEL_SCALED_RBF: np.array([0.1, 1.3, 25.0]),
        K.KERNEL_MATERN32: synthetic.HYP_MATERN32,
        K.KERNEL_ARD_RBF: np.array([2.33e-6, 0.0099, 300.0, 12.11, 33.75, 45.14]),
    }[kid]
    e = ExactGPEngine(kid, hyp)
    got = e.kernel_matrix(x1, x2)
    want = K.kernel_matrix(kid, hyp, x1, x2)
    e.close()
    assert got.shape == want.shape
    # elementwise: a few ulp of the entry, plus an absolute floor for deeply underflowing tails
    assert np.allclose(got, want, rtol=2e-13, atol=1e-300 + 1e-16 * np.abs(want).max())


def test_kernel_matrix_generic_dims():
    rng = np.random.default_rng(0)
    for d in (2, 3, 6):
        x1 = np.column_stack([np.sort(rng.uniform(0, 9, 50))] + [rng.normal(size=50) for _ in range(d - 1)])
        hyp = np.array([0.1, 10.0, 3.0] + [2.0 + 0.3 * i for i in range(d - 1)])
        e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
        got = e.kernel_matrix(x1)
        e.close()
        assert np.allclose(got, K.kernel_matrix(K.KERNEL_BATTGP, hyp, x1), rtol=2e-13, atol=1e-300)


# ---------------------------------------------------------------------------------------------
# fit / predict against the oracle
# ---------------------------------------------------------------------------------------------
def _check_case(kid, hyp, x, y, xq, rel=REL):
    gp = OracleGP(kid, hyp, x, y).fit()
    m_ref, v_ref = gp.predict(xq, clamp=False)
    e = ExactGPEngine(kid, hyp)
    lml = e.fit(x, y)
    m, v = e.predict(xq, min_var=-1.0)
    m_only = e.predict(xq, want_var=False)
    alpha = e.alpha()
    e.close()
    assert e.jitter == gp.jitter == 0.0
    assert abs(lml - gp.lml) <= rel * abs(gp.lml), (lml, gp.lml)
    assert rel_err(m, m_ref) < rel
    assert rel_err(m_only, m) < 1e-9  # mean-only path uses K_*X alpha, the variance path V^T z
    assert rel_err(alpha, gp.alpha) < 1e-4  # alpha is cond-amplified; the mean is the contract
    # variance: a catastrophic-cancellation quantity - compare against the prior scale
    kdiag = K.kernel_diag(kid, hyp, xq)
    assert np.max(np.abs(v - v_ref) / kdiag) < 1e-9, np.max(np.abs(v - v_ref) / kdiag)
    assert rel_err(v, v_ref) < 1e-5


@pytest.mark.parametrize("n", [1, 2, 10, 64, 65, 512, 1000, 2048, 3001])
def test_fit_predict_battgp_production_hyp(n):
    x, y = synthetic.make_cell_data(n, seed=77 + n)
    _check_case(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y, synthetic.make_query(x, 300 if n > 64 else 37))


@pytest.mark.parametrize("kid", [K.KERNEL_SCALED_RBF, K.KERNEL_MATERN32, K.KERNEL_ARD_RBF])
@pytest.mark.parametrize("n", [10, 700, 2048])
def test_fit_predict_other_kernels(kid, n):
    x, y = synthetic.make_cell_data(n, seed=n)
    xq = synthetic.make_query(x, 123)
    if kid == K.KERNEL_SCALED_RBF:
        mu, sd = x.mean(axis=0), x.std(axis=0)
        x, xq = (x - mu) / sd, (xq - mu) / sd
        hyp = np.array([2.33e-6, 0.0099, 1.5])
        y = y - y.mean()
    elif kid == K.KERNEL_MATERN32:
        hyp = synthetic.HYP_MATERN32
    else:
        hyp = np.array([2.33e-6, 0.0099, 300.0, 12.11, 33.75, 45.14])
    _check_case(kid, hyp, x, y, xq)


def test_golden_oracle_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_cases.npz"))
    for kname in ("k0", "k1", "k2", "k3"):
        for n in (1, 2, 10, 64, 512):
            p = f"{kname}_n{n}_"
            e = ExactGPEngine(int(g[p + "kernel_id"]), g[p + "hyp"])
            lml = e.fit(g[p + "x"], g[p + "y"])
            m, v = e.predict(g[p + "xq"], min_var=-1.0)
            m8, c8 = e.predict_cov(g[p + "xq"][:8])
            e.close()
            assert abs(lml - float(g[p + "lml"])) <= REL * abs(float(g[p + "lml"])), (kname, n)
            assert rel_err(m, g[p + "mean"]) < REL, (kname, n)
            scale = np.abs(g[p + "cov8"]).max() + 1e-300
            assert np.max(np.abs(c8 - g[p + "cov8"])) / scale < 1e-6, (kname, n)
            assert np.allclose(m8, m[:8], rtol=1e-12, atol=0)
            assert np.max(np.abs(v - g[p + "var"])) <= 1e-9 * K.kernel_diag(int(g[p + "kernel_id"]), g[p + "hyp"], g[p + "xq"]).max()


def test_golden_n2048_checksums(golden_dir):
    with open(os.path.join(golden_dir, "oracle_n2048.json")) as f:
        ref = json.load(f)
    x, y = synthetic.make_cell_data(2048)
    xq = synthetic.make_query(x)
    for name, kid, hyp in (("k0", K.KERNEL_BATTGP, synthetic.HYP_BATTGP), ("k2", K.KERNEL_MATERN32, synthetic.HYP_MATERN32)):
        e = ExactGPEngine(kid, hyp)
        lml = e.fit(x, y)
        m, v = e.predict(xq, min_var=-1.0)
        r = e.residuals(128)
        e.close()
        assert abs(lml - ref[name]["lml"]) <= REL * abs(ref[name]["lml"])
        assert abs(m.sum() - ref[name]["mean_sum"]) <= REL * abs(ref[name]["mean_sum"])
        assert np.allclose(m[:4], ref[name]["mean_first"], rtol=REL)
        assert abs(v.sum() - ref[name]["var_sum"]) <= 1e-5 * abs(ref[name]["var_sum"])
        assert r[0] < 1e-7 and r[1] < 1e-12, r


def test_stgp_egp_golden_on_gpu(golden_dir):
    """The reference's own cross-check (tests/gp/test_spatiotemporal_gp.py:218-282): exact GP with
    Wiener+ARD-RBF == Kalman stGP at 1e-6 rel, the Kalman side driven by the reference's (A, Q)."""
    g = np.load(os.path.join(golden_dir, "stgp_egp.npz"))
    xt, yt, sq, tt, hyp = g["xt"], g["yt"], g["sq"], g["tt"], g["hyp"]
    for i in range(len(tt)):
        e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
        e.fit(xt[: i + 1], yt[: i + 1])
        xq = np.hstack((np.full((sq.shape[0], 1), tt[i]), sq))
        m, v = e.predict(xq)
        e.close()
        assert np.linalg.norm(m - g["kalman_mean"][i]) < 1e-6 * np.linalg.norm(m)
        assert np.linalg.norm(v - g["kalman_var"][i]) < 1e-6 * np.linalg.norm(v)


def test_reference_known_answers_on_gpu():
    # tests/gp/test_standard_models.py:17-47
    e = ExactGPEngine(K.KERNEL_SCALED_RBF, [3.0, 3.0, 2.0])
    e.fit(np.array([[1.0]]), np.array([10.0]))
    m, v = e.predict(np.array([[1.0]]), min_var=-1.0)
    assert abs(m[0] - 5.0) < 1e-12 and abs(v[0] - 1.5) < 1e-12
    e.fit(np.array([[1.0], [1.0]]), np.array([10.0, 10.0]))
    m, v = e.predict(np.array([[1.0]]), min_var=-1.0)
    e.close()
    assert abs(m[0] - (5.0 / 1.5 + 10.0 / 3.0) / (1 / 1.5 + 1 / 3.0)) < 1e-12
    assert abs(v[0] - 1.0) < 1e-12


def test_variance_clamp_matches_gpytorch_min_variance():
    # duplicate training inputs with tiny noise => posterior variance ~ 0 => floored at 1e-10
    x = np.array([[1.0, 0.0], [1.0, 0.0], [2.0, 1.0]])
    y = np.array([1.0, 1.0, 2.0])
    hyp = np.array([1e-12, 1.0, 1.0, 1.0])
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    e.fit(x, y)
    _, v = e.predict(x)  # default min_var = 1e-10 (battcellgp_full.py:180 -> .variance)
    _, v_raw = e.predict(x, min_var=-1.0)
    e.close()
    assert np.all(v >= 1e-10) and np.any(v_raw < 1e-10)


def test_jitter_ladder_and_not_psd():
    # exactly singular in exact AND floating-point arithmetic: K = all-ones (t = 0 kills the
    # Wiener term, s_rbf = 1), zero noise => second pivot is exactly 1 - 1 = 0 everywhere
    x = np.zeros((3, 2))
    y = np.array([1.0, 1.0, 1.0])
    hyp = np.array([0.0, 1.0, 1.0, 1.0])
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        e.fit(x, y)
    assert e.jitter == 1e-8
    assert any(issubclass(i.category, NumericalWarning) for i in w)
    ref = OracleGP(K.KERNEL_BATTGP, hyp, x, y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.fit()
    assert ref.jitter == 1e-8 and abs(e.lml - ref.lml) < 1e-6 * abs(ref.lml)
    e.close()
    # hopeless: y irrelevant, K = -like via zero max_tries on a singular matrix
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    e.set_options(max_tries=0)
    with pytest.raises(NotPSDError):
        e.fit(x, y)
    e.close()


def test_refit_equals_fresh_fit():
    x, y = synthetic.make_cell_data(900, seed=3)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.fit(x, y)
    hyp2 = synthetic.HYP_BATTGP * np.array([2.0, 0.5, 1.5, 0.7, 1.2, 0.9])
    lml2 = e.refit(hyp2)
    e.close()
    assert abs(lml2 - OracleGP(K.KERNEL_BATTGP, hyp2, x, y).fit().lml) < REL * abs(lml2)


def test_option_nb_outer_does_not_change_results():
    x, y = synthetic.make_cell_data(1500, seed=9)
    xq = synthetic.make_query(x, 50)
    out = []
    for nb in (64, 256, 512, 1024):
        e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
        e.set_options(nb_outer=nb)
        lml = e.fit(x, y)
        m, v = e.predict(xq)
        e.close()
        out.append((lml, m, v))
    for lml, m, v in out[1:]:
        assert abs(lml - out[0][0]) < 1e-9 * abs(lml)
        assert rel_err(m, out[0][1]) < 1e-9


def test_size_independent_properties_medium_n():
    """At sizes the oracle would take too long for: residual checks computed on the device and
    linearity of the posterior mean in y."""
    n = 8192
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.fit(x, y)
    m1 = e.predict(xq, want_var=False)
    r = e.residuals(512)
    assert r[0] < 1e-6 and r[1] < 1e-11, r
    e.fit(x, 3.0 * y)
    m3 = e.predict(xq, want_var=False)
    e.close()
    assert rel_err(m3, 3.0 * m1) < 1e-9


@pytest.mark.parametrize("n,m", [(1, 1), (65, 300), (700, 37), (2048, 300), (300, 1000)])
def test_fit_predict_fused_equals_separate(n, m):
    x, y = synthetic.make_cell_data(n, seed=n + m)
    xq = synthetic.make_query(x, m)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    lml_f, mean_f, var_f = e.fit_predict(x, y, xq, min_var=-1.0)
    # the handle stays fitted: another query set goes through the stored factor
    xq2 = synthetic.make_query(x, 11, op=(-30.0, 60.0, 20.0))
    m2, v2 = e.predict(xq2, min_var=-1.0)
    e.close()
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    lml_s = e.fit(x, y)
    mean_s, var_s = e.predict(xq, min_var=-1.0)
    m2s, v2s = e.predict(xq2, min_var=-1.0)
    e.close()
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq, clamp=False)
    assert abs(lml_f - lml_s) <= 1e-12 * abs(lml_s) and abs(lml_f - ref.lml) <= REL * abs(ref.lml)
    assert rel_err(mean_f, m_ref) < REL and rel_err(mean_f, mean_s) < 1e-9
    kd = K.kernel_diag(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, xq)
    assert np.max(np.abs(var_f - v_ref) / kd) < 1e-9 and np.max(np.abs(var_f - var_s) / kd) < 1e-10
    assert rel_err(m2, m2s) < 1e-9 and np.max(np.abs(v2 - v2s)) < 1e-10 * synthetic.OUTPUTSCALE_RBF


def test_fit_predict_with_jitter_retry():
    x = np.zeros((3, 2))
    y = np.ones(3)
    hyp = np.array([0.0, 1.0, 1.0, 1.0])
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        lml, mean, var = e.fit_predict(x, y, np.array([[0.0, 0.0], [1.0, 0.5]]), min_var=-1.0)
    assert e.jitter == 1e-8
    e.close()
    ref = OracleGP(K.KERNEL_BATTGP, hyp, x, y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref.fit()
    m_ref, v_ref = ref.predict(np.array([[0.0, 0.0], [1.0, 0.5]]), clamp=False)
    assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
    assert np.allclose(mean, m_ref, rtol=1e-6) and np.allclose(var, v_ref, rtol=1e-5, atol=1e-9)



@pytest.mark.parametrize("d", [2, 3, 6])
def test_fit_predict_generic_input_dimension(d):
    """The reference's kernel builder takes any number of RBF dims (tests/gp/test_spatiotemporal_gp.py:18-39);
    D = 4 has a specialised fill kernel, every other D goes through the generic one."""
    rng = np.random.default_rng(d)
    n = 333
    x = np.column_stack([np.sort(rng.uniform(0, 9, n))] + [rng.normal(size=n) for _ in range(d - 1)])
    y = np.sin(x[:, 0]) + 0.1 * rng.normal(size=n)
    xq = np.column_stack([rng.uniform(0, 9, 40)] + [rng.normal(size=40) for _ in range(d - 1)])
    hyp = np.array([0.1, 10.0, 3.0] + [2.0 + 0.3 * i for i in range(d - 1)])
    _check_case(K.KERNEL_BATTGP, hyp, x, y, xq)
    if d == 3:
        _check_case(K.KERNEL_MATERN32, np.array([0.05, 1.5, 2.0, 1.0, 3.0]), x, y, xq)
        _check_case(K.KERNEL_SCALED_RBF, np.array([0.05, 1.5, 2.0]), x[:, :1], y, xq[:, :1])


def test_errors_are_reported_not_crashed():
    from battgp_amd.engine import EngineError

    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    x, y = synthetic.make_cell_data(100)
    with pytest.raises(EngineError, match="no successful fit"):
        e.predict(x[:3])
    with pytest.raises(EngineError, match="expects 6 hyper-parameters|hyper-parameters"):
        e.fit(x[:, :3], y)  # D = 3 with a 6-entry hyp vector
    with pytest.raises(EngineError, match="not a positive finite"):
        e.set_hyp([1e-6, -1.0, 1.0, 1.0, 1.0, 1.0])
    with pytest.raises(EngineError, match="nb_outer"):
        e.set_options(nb_outer=100)
    # NaN in the targets is harmless for the factorisation; NaN in the inputs poisons Sigma -> NotPSD
    e.set_hyp(synthetic.HYP_BATTGP)
    xb = x.copy()
    xb[7, 2] = np.nan
    with pytest.raises(NotPSDError):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e.fit(xb, y)
    e.fit(x, y)  # the handle is still usable afterwards
    assert np.isfinite(e.lml)
    e.close()


def test_allocation_failure_is_an_error_message():
    from battgp_amd.engine import EngineError

    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    n = 400_000  # 1.28 TB: cannot fit
    x = np.zeros((n, 4))
    with pytest.raises(EngineError, match="hipMalloc"):
        e.fit(x, np.zeros(n))
    x, y = synthetic.make_cell_data(200)
    assert np.isfinite(e.fit(x, y))
    e.close()


@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_MATERN32])
@pytest.mark.parametrize("n,nb", [(700, 256), (2048, 512), (3001, 512), (5000, 1024), (4100, 2048)])
def test_panel_schemes_agree_and_match_oracle(kid, n, nb):
    """scheme 1 (chain on the diagonal block + one TRSM-by-explicit-inverse GEMM) against scheme 0 (64-wide
    chain over all rows) and the oracle: same algebra, different rounding"""
    hyp = synthetic.HYP_BATTGP if kid == K.KERNEL_BATTGP else synthetic.HYP_MATERN32
    x, y = synthetic.make_cell_data(n, seed=n)
    xq = synthetic.make_query(x, 200)
    out = []
    for scheme in (0, 1):
        e = ExactGPEngine(kid, hyp)
        e.set_options(nb_outer=nb)
        e.set_panel_scheme(scheme)
        lml, m, v = e.fit_predict(x, y, xq, min_var=-1.0)
        m2, v2 = e.predict(xq, min_var=-1.0)
        res = e.residuals(256)
        e.close()
        assert rel_err(m2, m) < 1e-8
        assert res[0] < 1e-6 and res[1] < 1e-11, res
        out.append((lml, m, v))
    assert abs(out[0][0] - out[1][0]) <= 1e-9 * abs(out[0][0])
    assert rel_err(out[1][1], out[0][1]) < 1e-8
    gp = OracleGP(kid, hyp, x, y).fit()
    m_ref, v_ref = gp.predict(xq, clamp=False)
    assert abs(out[1][0] - gp.lml) <= REL * abs(gp.lml)
    assert rel_err(out[1][1], m_ref) < REL
    assert np.max(np.abs(out[1][2] - v_ref) / K.kernel_diag(kid, hyp, xq)) < 1e-9


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(problems(n_max=3000, m_max=400, noise_lo=-3.0, la_words=LA_MEASURED))
def test_random_problems_match_the_oracle(prob):
    """the property of tests/test_emu_property.py at sizes with dozens of panels: ragged N / M, D = 1..6, all kernels,
    random hyper-parameters, duplicated points, both panel schemes, the look-ahead words that have run on the GPU, slab
    layout - LML and posterior against the oracle; the gradient of the same problems in tests/test_gpu_grad.py (the
    optional schedules built while the GPU pool was closed take the property in tests/test_gpu_zz_optional_schedules.py,
    in a child process)"""
    check_problem(*prob)


@pytest.mark.gpu_sized
def test_default_panel_width_is_chosen_by_size():
    """no explicit nb_outer: 512 below N = 32 768, 1024 from there on; results agree with an explicit 512"""
    n = 33000
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    lml_auto, m_auto, _ = e.fit_predict(x, y, xq)
    launches_auto = e.phase_times()["trail_launches"]
    e.set_options(nb_outer=512)
    lml_512, m_512, _ = e.fit_predict(x, y, xq)
    launches_512 = e.phase_times()["trail_launches"]
    r = e.residuals(256)
    e.close()
    assert launches_auto < 0.6 * launches_512
    assert abs(lml_auto - lml_512) <= 1e-9 * abs(lml_512)
    assert rel_err(m_auto, m_512) < 1e-8
    assert r[0] < 1e-6 and r[1] < 1e-11


@pytest.mark.parametrize("scheme", [0, 1])
def test_lookahead_depth_does_not_change_a_single_bit(scheme):
    """look-ahead depth d: the panel stream updates the next panel left-looking from the last d panels, the
    main stream the panels from k+d+1 on - every element still receives its updates in panel order"""
    n = 5000
    x, y = synthetic.make_cell_data(n, seed=1)
    xq = synthetic.make_query(x, 100)
    out = []
    for la in (0, 1, 2, 3, 4, 1 | 8, 2 | 8):
        e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
        e.set_options(nb_outer=512, lookahead=la)
        e.set_panel_scheme(scheme)
        lml, m, v = e.fit_predict(x, y, xq, min_var=-1.0)
        out.append((lml, m, v, e.alpha()))
        e.close()
    for lml, m, v, a in out[1:]:
        assert lml == out[0][0]
        assert np.array_equal(m, out[0][1]) and np.array_equal(v, out[0][2]) and np.array_equal(a, out[0][3])


def test_unsorted_time_column_takes_the_general_wiener_path():
    """The training fill uses min(t_i, t_j) = t_j below the diagonal only when the time column is ascending (checked on
    the device at upload); shuffled rows must give the same GP (LML, posterior) through the general formula."""
    n = 1700
    x, y = synthetic.make_cell_data(n, seed=77)
    xq = synthetic.make_query(x, 40)
    perm = np.random.default_rng(5).permutation(n)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq, clamp=False)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    for xs, ys in ((x, y), (x[perm], y[perm])):
        lml, mean, var = e.fit_predict(xs, ys, xq, min_var=-1.0)
        assert abs(lml - ref.lml) <= REL * abs(ref.lml)
        assert rel_err(mean, m_ref) <= REL
        assert np.max(np.abs(var - v_ref)) <= 1e-9 * synthetic.OUTPUTSCALE_RBF
    # ties in the time column are "sorted" too: both formulas agree on them
    xt = x.copy()
    xt[100:110, 0] = xt[100, 0]
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, xt, y).fit()
    assert abs(e.fit(xt, y) - ref.lml) <= REL * abs(ref.lml)
    e.close()

