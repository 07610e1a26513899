"""Column-slab layout of the factor (bgp_set_layout): ~4 N (N + W) bytes instead of 8 N^2, the layout
that lets N = 262 144 live on one MI355X ("N_max per GPU" of BASELINE.json's metric).  Every element
receives exactly the same arithmetic as in the full-square layout, so the results must be
BIT-IDENTICAL to it - and within the north-star tolerance of the CPU oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import EngineError, ExactGPEngine, trim_pool  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402

REL = 1e-6


def rel_err(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _hyp(kid):
    if kid == K.KERNEL_BATTGP:
        return synthetic.HYP_BATTGP
    if kid == K.KERNEL_MATERN32:
        return synthetic.HYP_MATERN32
    if kid == K.KERNEL_SCALED_RBF:
        return np.array([2.33e-6, 0.0099, 150.0])
    return np.array([2.33e-6, 0.0099, 300.0, 12.11, 33.75, 45.14])


def _run(kid, x, y, xq, slab, nb=None, fused=False):
    e = ExactGPEngine(kid, _hyp(kid))
    if nb:
        e.set_options(nb_outer=nb)
    e.set_layout(slab)
    if fused:
        lml, m, v = e.fit_predict(x, y, xq, min_var=-1.0)
    else:
        lml = e.fit(x, y)
        m, v = e.predict(xq, min_var=-1.0)
    width, nbytes = e.layout()
    alpha = e.alpha()
    res = e.residuals(128)
    e.close()
    return dict(lml=lml, m=m, v=v, alpha=alpha, res=res, width=width, nbytes=nbytes)


@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_SCALED_RBF, K.KERNEL_MATERN32, K.KERNEL_ARD_RBF])
@pytest.mark.parametrize("n,slab,nb", [(1000, 512, 512), (3001, 512, 256), (3001, 1024, 512), (2048, 1024, 1024), (2500, 2048, 512)])
@pytest.mark.parametrize("fused", [False, True])
def test_slab_layout_bit_identical_to_full_square(kid, n, slab, nb, fused):
    x, y = synthetic.make_cell_data(n, seed=n + kid)
    xq = synthetic.make_query(x, 300)
    full = _run(kid, x, y, xq, -1, nb, fused)
    slabbed = _run(kid, x, y, xq, slab, nb, fused)
    assert full["width"] == 0
    assert slabbed["width"] == slab
    assert slabbed["nbytes"] < full["nbytes"]
    assert slabbed["lml"] == full["lml"]
    assert np.array_equal(slabbed["m"], full["m"])
    assert np.array_equal(slabbed["v"], full["v"])
    assert np.array_equal(slabbed["alpha"], full["alpha"])
    assert slabbed["res"] == full["res"]


@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_MATERN32])
def test_slab_layout_matches_oracle(kid):
    n = 3001
    x, y = synthetic.make_cell_data(n, seed=5)
    xq = synthetic.make_query(x, 123)
    gp = OracleGP(kid, _hyp(kid), x, y).fit()
    m_ref, v_ref = gp.predict(xq, clamp=False)
    for fused in (False, True):
        got = _run(kid, x, y, xq, 512, fused=fused)
        assert abs(got["lml"] - gp.lml) <= REL * abs(gp.lml)
        assert rel_err(got["m"], m_ref) < REL
        assert np.max(np.abs(got["v"] - v_ref) / K.kernel_diag(kid, _hyp(kid), xq)) < 1e-9
        assert got["res"][0] < 1e-6 and got["res"][1] < 1e-11


def test_slab_layout_saves_the_upper_triangle():
    n = 8192
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x)
    full = _run(K.KERNEL_BATTGP, x, y, xq, -1)
    slabbed = _run(K.KERNEL_BATTGP, x, y, xq, 512)
    assert slabbed["nbytes"] < 0.56 * full["nbytes"]  # 4 N (N + W) + riding rows vs 8 N^2
    assert slabbed["lml"] == full["lml"] and np.array_equal(slabbed["m"], full["m"])


def test_slab_layout_refit_and_cov():
    n = 1500
    x, y = synthetic.make_cell_data(n, seed=11)
    xq = synthetic.make_query(x, 40)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.set_layout(512)
    e.fit(x, y)
    hyp2 = synthetic.HYP_BATTGP * np.array([2.0, 0.5, 1.5, 0.7, 1.2, 0.9])
    lml2 = e.refit(hyp2)
    assert e.layout()[0] == 512
    gp2 = OracleGP(K.KERNEL_BATTGP, hyp2, x, y).fit()
    assert abs(lml2 - gp2.lml) <= REL * abs(gp2.lml)
    mean, cov = e.predict_cov(xq)
    m_ref, c_ref = gp2.predict(xq, full_cov=True)
    assert rel_err(mean, m_ref) < REL
    assert np.max(np.abs(cov - c_ref)) < 1e-9 * hyp2[2]
    e.close()


def test_slab_layout_jitter_ladder():
    """duplicate points + zero noise: the ladder must run (and land on the same rung) in both layouts"""
    import warnings

    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, size=(700, 4)) * np.array([100.0, 10.0, 10.0, 10.0])
    x[300:] = x[:400]
    y = rng.normal(size=700)
    hyp = np.array([0.0, 1e-9, 1.0, 3.0, 3.0, 3.0])
    out = []
    for slab in (-1, 512):
        e = ExactGPEngine(K.KERNEL_BATTGP, hyp)
        e.set_layout(slab)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                lml = e.fit(x, y)
                out.append((lml, e.jitter))
            except Exception as exc:  # both layouts must fail the same way too
                out.append((type(exc).__name__, None))
        e.close()
    assert out[0] == out[1], out


def test_slab_width_must_match_the_panel_width():
    x, y = synthetic.make_cell_data(2000)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.set_options(nb_outer=512)
    e.set_layout(768)
    with pytest.raises(EngineError, match="multiple of nb_outer"):
        e.fit(x, y)
    with pytest.raises(EngineError):
        e.set_layout(100)
    e.set_layout(1024)
    assert np.isfinite(e.fit(x, y))
    assert e.layout()[0] == 1024
    e.set_layout(0)  # automatic: plenty of HBM -> full square
    assert np.isfinite(e.fit(x, y))
    assert e.layout()[0] == 0
    e.close()


@pytest.mark.gpu_sized
def test_auto_layout_switches_to_slabs_when_the_square_does_not_fit():
    """occupy HBM with a torch tensor so that an N = 20 000 square (3.2 GB + margin) no longer fits"""
    n = 20000
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x)
    ref = _run(K.KERNEL_BATTGP, x, y, xq, -1, fused=True)
    trim_pool()  # parked handles would otherwise hold the square this test wants to be too big
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    leave = int(3.6e9)  # full square needs 3.3 GB + 0.8 GB margin; slabs of 8192 need 2.3 GB + margin
    hog = torch.empty(free - leave, dtype=torch.uint8, device="cuda")
    try:
        e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
        lml, m, v = e.fit_predict(x, y, xq, min_var=-1.0)
        width, nbytes = e.layout()
        e.close()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert width > 0 and nbytes < ref["nbytes"]
    assert lml == ref["lml"] and np.array_equal(m, ref["m"]) and np.array_equal(v, ref["v"])
