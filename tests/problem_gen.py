"""Random exact-GP problems and the oracle check behind the property tests (tests/test_emu_property.py on the CPU build
of the kernel sources, tests/test_gpu_parity.py::test_random_problems_match_the_oracle on the GPU): ragged sizes, input
dimensions 1..6, all four kernels, hyper-parameters over decades, unsorted / duplicated time stamps, coincident points,
both panel schemes, every look-ahead word, the column-slab layout."""

import warnings

import numpy as np
from hypothesis import strategies as st


LA_MEASURED = [0, 1, 2, 1 | 8]  # look-ahead words whose schedules have run (and been timed) on the GPU
LA_OPTIONAL = [1 | 32, 1 | 32 | 64, 1 | 128, 1 | 32 | 64 | 128]  # slim chain kernels, split panels, fused update + tile Cholesky: experimental library only


@st.composite
def problems(draw, n_max=170, m_max=60, noise_lo=-5.0, la_words=LA_MEASURED):
    kid = draw(st.sampled_from([0, 1, 2, 3]))
    d = draw(st.integers(2, 6)) if kid == 0 else draw(st.integers(1, 6))
    n = draw(st.integers(1, n_max))
    m = draw(st.integers(1, m_max))
    seed = draw(st.integers(0, 2**31 - 1))
    log = lambda lo, hi: 10.0 ** draw(st.floats(lo, hi, allow_nan=False, allow_infinity=False))  # noqa: E731
    noise = log(noise_lo, -1)  # (with the output scales below: condition numbers up to ~ n 10^(1 - noise_lo))
    if kid == 0:  # sigma^2, s_wiener, s_rbf, l_1..l_{d-1}
        hyp = [noise, log(-6, -2), log(-2, 1)] + [log(-0.5, 1) for _ in range(d - 1)]
    elif kid == 1:  # sigma^2, s, l
        hyp = [noise, log(-2, 1), log(-0.5, 1)]
    else:  # sigma^2, s, l_1..l_d
        hyp = [noise, log(-2, 1)] + [log(-0.5, 1) for _ in range(d)]
    opts = dict(
        nb=draw(st.sampled_from([64, 128])), scheme=draw(st.sampled_from([0, 1])),
        la=draw(st.sampled_from(la_words)), slab=draw(st.sampled_from([0, 0, 128])),
        sort_time=draw(st.booleans()), dup=draw(st.booleans()), fused=draw(st.booleans()),
    )
    return kid, d, n, m, seed, np.array(hyp), opts


def make_data(kid, d, n, m, seed, opts):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, d)) * 2.0
    if kid == 0:
        x[:, 0] = rng.uniform(0.0, 30.0, n)  # times >= 0 (the integrated Wiener kernel lives on t >= 0)
        if opts["sort_time"]:
            x = x[np.argsort(x[:, 0])]
    if opts["dup"] and n >= 4:
        x[n // 2] = x[n // 2 - 1]      # two coincident points: Sigma is singular up to the noise
        x[-1, 0] = x[0, 0]             # and a repeated first coordinate
    y = np.sin(x[:, 0]) + 0.3 * rng.normal(size=n)
    xq = rng.normal(size=(m, d)) * 2.0
    if kid == 0:
        xq[:, 0] = rng.uniform(0.0, 35.0, m)
    if m >= 2:
        xq[0] = x[0]  # a query ON a training point
    return np.ascontiguousarray(x), y, np.ascontiguousarray(xq)


def check_problem(kid, d, n, m, seed, hyp, opts, tol_lml=1e-6, grad=False):
    """LML and posterior against the oracle; ``grad=True`` adds the analytic gradient (the GPU suite asks for it only in
    tests/test_gpu_grad.py, which runs behind the fit / predict modules)."""
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    x, y, xq = make_data(kid, d, n, m, seed, opts)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = OracleGP(kid, hyp, x, y).fit()
        m_ref, v_ref = ref.predict(xq, clamp=False)
        e = ExactGPEngine(kid, hyp)
        try:
            e.set_options(nb_outer=opts["nb"], lookahead=opts["la"])
            e.set_panel_scheme(opts["scheme"])
            if opts["slab"]:
                e.set_layout(opts["slab"])
            if opts["fused"]:
                lml, mean, var = e.fit_predict(x, y, xq, min_var=-1.0)
            else:
                lml = e.fit(x, y)
                mean, var = e.predict(xq, min_var=-1.0)
            assert e.jitter == ref.jitter
            scale = K.kernel_diag(kid, hyp, xq)
            # the tolerance of north_star (1e-6 relative on LML and mean), the LML with an absolute floor where it
            # passes through zero
            assert abs(lml - ref.lml) <= tol_lml * max(abs(ref.lml), 1.0), (lml, ref.lml)
            assert np.linalg.norm(mean - m_ref) <= 1e-6 * max(np.linalg.norm(m_ref), 1e-3 * np.sqrt(m)), (mean[:3], m_ref[:3])
            assert np.max(np.abs(var - v_ref) / scale) < 1e-7
            if grad and n >= 2:
                g = e.lml_grad()
                _, g_ref = lml_and_grad(kid, hyp, x, y)
                assert np.all(np.abs(g - g_ref) <= 1e-5 * np.maximum(np.abs(g_ref), 1e-3 * np.max(np.abs(g_ref)))), (g, g_ref)
        finally:
            e.close()
