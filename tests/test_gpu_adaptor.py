"""GPU tests of the reference-shaped surface (BatteryCellGP_Full, ScaledRBFModel, training loops)
- written to read like the reference's own tests (tests/gp/test_standard_models.py,
tests/gp/test_spatiotemporal_gp.py:218-282), with the oracle as the checker."""

import gc
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from battgp_amd import config as cfg  # noqa: E402
from battgp_amd import synthetic, training  # noqa: E402
from battgp_amd.battcellgp_full import BatteryCellGP_Full, build_cellmodel_full, predict_cells_concurrently  # noqa: E402
from battgp_amd.operating_point import Op  # noqa: E402
from battgp_amd.standard_models import ScaledRBFModel  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402


class TestStandardModels:
    """Mirror of the reference's tests/gp/test_standard_models.py:12-47."""

    def test_rbf_1d(self):
        x = np.array([[1.0]])
        gp = ScaledRBFModel(torch.Tensor(x), torch.Tensor([10.0]), noise_variance=3.0, outputscale=3.0, lengthscale=2.0)
        (yq, var_yq) = gp.predict(x)
        assert yq[0] == pytest.approx(5.0, abs=1e-7)
        assert var_yq[0] == pytest.approx(1.5, abs=1e-5)
        gp = ScaledRBFModel(torch.Tensor([[1.0], [1.0]]), torch.Tensor([10.0, 10.0]), noise_variance=3.0, outputscale=3.0, lengthscale=2.0)
        (yq, var_yq) = gp.predict(x)
        assert yq[0] == pytest.approx((5.0 / 1.5 + 10.0 / 3.0) / (1 / 1.5 + 1 / 3.0), abs=1e-5)
        assert var_yq[0] == pytest.approx(3.0 / 3, abs=1e-5)

    def test_noise_free_closed_forms(self):
        # tests/gp/test_recursive_gp.py:85-102: mean = 10 exp(-d^2/8), var = 3 - 3 exp(-d^2/4)
        gp = ScaledRBFModel(np.array([[1.0]]), np.array([10.0]), noise_variance=1e-300, outputscale=3.0, lengthscale=2.0)
        for d in (0.0, 0.5, 1.0, 3.0):
            m, v = gp.predict(np.array([[1.0 + d]]))
            assert m[0] == pytest.approx(10.0 * np.exp(-d * d / 8.0), abs=1e-10)
            assert v[0] == pytest.approx(3.0 - 3.0 * np.exp(-d * d / 4.0), abs=1e-10)

    def test_full_covariance(self):
        # the exact-GP side of tests/gp/test_recursive_gp.py:195-232 (50 train, 50 query, 3-D)
        rng = np.random.default_rng(1)
        xt, yt, xq = rng.uniform(-5, 5, (50, 3)), rng.normal(size=50), rng.uniform(-5, 5, (50, 3))
        gp = ScaledRBFModel(xt, yt, noise_variance=3.0, outputscale=3.0, lengthscale=2.0)
        m, c = gp.predict(xq, full_cov=True)
        ref = OracleGP(K.KERNEL_SCALED_RBF, [3.0, 3.0, 2.0], xt, yt).fit()
        m_ref, c_ref = ref.predict(xq, full_cov=True)
        assert np.linalg.norm(m - m_ref) < 1e-9 * np.linalg.norm(m_ref)
        assert np.linalg.norm(c - c_ref) < 1e-9 * np.linalg.norm(c_ref)
        assert np.allclose(c, c.T, rtol=0, atol=1e-12)


def test_cell_model_matches_oracle_and_reference_output_format():
    x, y = synthetic.make_cell_data(1200, seed=5)
    cell = BatteryCellGP_Full(x, y, cellnr=4, device=0)
    t = np.linspace(x[0, 0], x[-1, 0], 300)
    op = Op(*synthetic.REF_OP)
    df = cell.predict_r0_op(op, t)
    assert list(df.columns) == ["t", "r0_acausal_c4", "r0var_acausal_c4"]
    xq = np.column_stack((t, np.full(300, op.I), np.full(300, op.SOC), np.full(300, op.T)))
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)  # clamped at 1e-10 like .variance
    assert np.linalg.norm(df["r0_acausal_c4"] - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    assert np.max(np.abs(df["r0var_acausal_c4"] - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    # predict(): tuple / no_cov / full_cov semantics of battcellgp_full.py:168-195
    m_only = cell.predict(xq, no_cov=True)
    assert isinstance(m_only, np.ndarray) and np.allclose(m_only, df["r0_acausal_c4"], rtol=1e-12)
    m, cov = cell.predict(xq[:5], full_cov=True)
    assert cov.shape == (5, 5) and np.isnan(cov[0, 1]) and np.allclose(np.diag(cov), df["r0var_acausal_c4"][:5])
    # pack model tag
    pack = BatteryCellGP_Full(x, y, cellnr=-1, device=torch.device("cuda", 0))
    assert list(pack.predict_r0_op(op, t[:3]).columns) == ["t", "r0_acausal_pack", "r0var_acausal_pack"]
    # destroy like BattGP_Full.predict_cell_r0_op does
    del cell.model
    gc.collect()
    torch.cuda.empty_cache()


def test_model_object_surface_used_by_callers():
    x, y = synthetic.make_cell_data(300, seed=2)
    cell = BatteryCellGP_Full(x, y, 1, device=0)
    mdl = cell.model
    assert mdl.train_inputs[0].is_cuda and tuple(mdl.train_targets.shape) == (300,)
    t0 = mdl.train_inputs[0][:, 0]
    assert float(t0[0].detach().cpu()) == x[0, 0]  # battgp_full.py:84,98
    out = mdl(torch.tensor(x[:7]))
    assert out.mean.shape == (7,) and out.variance.shape == (7,) and bool((out.variance >= 1e-10).all())
    assert float(mdl.noise_variance) == cfg.NOISE_VARIANCE[0]
    assert tuple(mdl.lengthscale_rbf.detach().cpu().numpy()[0]) == cfg.LENGTHSCALE_RBF
    assert len(list(mdl.parameters())) == 4


class _FakeBattData:
    """Only the contract build_cellmodel_full needs (src/batt_data/batt_data.py:180-256)."""

    cell_nrs = [1, 2, 3]
    age = 1200.0

    def generateTrainingData(self, cellnr, max_training_data, max_age):
        return synthetic.make_cell_data(max_training_data, seed=500 + cellnr)


def test_battgp_full_call_sequence():
    """The call sequence of BattGP_Full (battgp_full.py:41-125) on the plugin: pack + cells, shared 300-point
    grid from cellmodels[0].model.train_inputs, predict_r0_op each, merge on t, free models."""
    bd = _FakeBattData()
    pack = build_cellmodel_full(-1, bd, max_training_data=400, device=0)
    cells = [build_cellmodel_full(c, bd, max_training_data=400, device=0) for c in bd.cell_nrs]
    t = cells[0].model.train_inputs[0][:, 0]
    tt = np.linspace(t[0].detach().cpu(), bd.age, 300)
    op = Op(*synthetic.REF_OP)
    df = pack.predict_r0_op(op=op, t=tt)
    del pack.model
    for i, c in enumerate(cells):
        df = df.merge(c.predict_r0_op(op=op, t=tt))
        if i < len(cells) - 1:
            del c.model
    assert df.shape == (300, 1 + 2 * 4)
    assert {"r0_acausal_pack", "r0_acausal_c1", "r0var_acausal_c3"} <= set(df.columns)
    assert np.isfinite(df.to_numpy()).all()
    assert cells[-1].get_training_data()[0].shape == (400, 4)  # plotting.py:233,259 keeps the last model


def test_concurrent_models_from_threads():
    """battgp.py:191-216 drives cells from a thread pool: distinct handles must be independent."""
    bd = _FakeBattData()
    cells = [build_cellmodel_full(c, bd, max_training_data=600, device=0) for c in (1, 2, 3)]
    op = Op(*synthetic.REF_OP)
    tt = np.linspace(0.0, 1200.0, 100)
    serial = [c.predict_r0_op(op, tt) for c in cells]
    for c in cells:
        c.model._invalidate()
    out = [None] * 3

    def run(i):
        out[i] = cells[i].predict_r0_op(op, tt)

    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    for a, b in zip(serial, out):
        assert np.array_equal(a.to_numpy(), b.to_numpy())  # same fused path, run concurrently: bit-identical
    dfs = predict_cells_concurrently(cells, op, tt)  # models are fitted now: separate solve pass of the query block
    for a, b in zip(serial, dfs):
        assert np.allclose(a.to_numpy(), b.to_numpy(), rtol=1e-8, atol=1e-13)


def test_neg_mll_value_is_what_the_reference_reports():
    x, y = synthetic.make_cell_data(500, seed=11)
    cell = BatteryCellGP_Full(x, y, 1, device=0)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    n = len(y)
    assert cell.model.neg_mll() * n == pytest.approx(ref.neg_mll_scaled, rel=1e-6)  # training.py:43 loss*loss_scale


def test_train_hyperparameters_decreases_loss_and_updates_params(monkeypatch):
    x, y = synthetic.make_cell_data(400, seed=21)
    cell = BatteryCellGP_Full(x, y, 2, device=0, max_iter=15, lr=0.05, rel_tol=0.0,
                              noise_variance=(1e-5,), outputscale_rbf=0.05)
    before = cell.get_parameters()
    losses = cell.train_hyperparameters(messages=False)
    assert isinstance(losses, np.ndarray) and len(losses) == 16
    assert losses[-2] < losses[0]
    after = cell.get_parameters()
    assert after["noise_variance"] != before["noise_variance"][0]
    assert isinstance(after["lengthscale_rbf"], tuple) and len(after["lengthscale_rbf"]) == 3
    assert cell.marginallikelihood == losses[-1]
    # the model is usable after training and agrees with the oracle at the trained values
    hyp = cell.model.hyp_vector()
    xq = synthetic.make_query(x, 20)
    m = cell.predict(xq, no_cov=True)
    m_ref, _ = OracleGP(K.KERNEL_BATTGP, hyp, x, y).fit().predict(xq)
    assert np.linalg.norm(m - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    monkeypatch.setitem(cfg.HYPER_OPT_PARAMS, "opt_algorithm", "nope")
    with pytest.raises(ValueError, match="not implemented"):
        cell.train_hyperparameters(messages=False)


def test_analytic_raw_gradient_matches_finite_differences():
    x, y = synthetic.make_cell_data(300, seed=8)
    cell = BatteryCellGP_Full(x, y, 1, device=0, noise_variance=(1e-5,), outputscale_rbf=0.05)
    raw = cell.model.raw_vector()
    f, g = training._loss_and_grad(cell.model, raw)
    f_fd, g_fd = training._loss_and_grad_fd(cell.model, raw, rel_step=1e-5)
    assert f == pytest.approx(f_fd, rel=1e-12)
    assert np.allclose(g, g_fd, rtol=2e-4, atol=1e-8 * np.abs(g_fd).max()), (g, g_fd)


def test_system_driver_end_to_end(tmp_path):
    """BattGP_Full on a synthetic 3-cell system: the gp_runner full_gp call sequence
    (gp_runner.py:68-96), result frame layout, feather/json artefacts, get_cell_data."""
    import json

    import pandas as pd

    from battgp_amd.battgp_full import BattGP_Full
    from battgp_amd.synthetic import SyntheticBattData

    try:  # feather needs pyarrow; it is in the image, but do not let a box without it fail the parity part
        import pyarrow  # noqa: F401

        have_arrow = True
    except ImportError:
        have_arrow = False
    bd = SyntheticBattData("sys7", n_cells=3, seed=7)
    sysm = BattGP_Full(bd, max_training_data=500, device=0, save_path=str(tmp_path))
    res = sysm.predict_cell_r0_op(save=have_arrow)
    df = res.df
    assert df.shape == (300, 1 + 2 * 4)
    assert list(df.columns)[:3] == ["t", "r0_acausal_pack", "r0var_acausal_pack"]
    assert np.isfinite(df.to_numpy()).all() and (df.filter(like="r0var").to_numpy() >= 1e-10).all()
    # every cell column equals an oracle GP on that cell's data
    x, y = bd.generateTrainingData(2, 500)
    xq = np.column_stack((df["t"], np.full(300, res.ref_op.I), np.full(300, res.ref_op.SOC), np.full(300, res.ref_op.T)))
    m_ref, _ = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit().predict(xq)
    assert np.linalg.norm(df["r0_acausal_c2"] - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    # artefacts (battgp.py:233-262)
    if have_arrow:
        saved = pd.read_feather(tmp_path / "sys7" / "battgpf_df.feather")
        assert saved.equals(df)
        info = json.load(open(tmp_path / "sys7" / "battgpf_info.json"))
        assert info["ref_point"] == res.ref_op.disp_str()
    # BattGPResult.get_cell_data naming rules (battgp.py:23-92)
    one = res.get_cell_data(2, ["t", "r0", "r0var"])
    assert list(one.columns) == ["t", "r0", "r0var"]
    many = res.get_cell_data([1, 2, 3], ["t", "r0", "r0var"])
    assert list(many.columns) == ["t", "r0_c1", "r0_c2", "r0_c3", "r0var_c1", "r0var_c2", "r0var_c3"]
    with pytest.raises(ValueError, match="not available"):
        res.get_cell_data(1, ["dr0"])
    assert res.get_cell_data(1, ["t", "dr0"], missing_behaviour="ignore").shape[1] == 1
    # all models but the last were destroyed (battgp_full.py:102-120)
    assert not hasattr(sysm.packmodel, "model") and not hasattr(sysm.cellmodels[0], "model")
    assert hasattr(sysm.cellmodels[-1], "model")
    assert sysm.get_cell_model(-1) is sysm.packmodel and sysm.get_cell_model(3).cellnr == 3


def test_system_driver_concurrent_gps_bit_identical():
    """The nine GPs of a system run concurrently on the one GPU (one handle + stream set + host thread each) by default
    at the reference's sizes; the frame must be BIT-identical to the strictly sequential loop of the reference
    (src/batt_models/battgp_full.py:100-121)."""
    from battgp_amd.battgp_full import BattGP_Full
    from battgp_amd.synthetic import SyntheticBattData

    bd = SyntheticBattData("sysc", n_cells=8, seed=3)
    seq = BattGP_Full(bd, max_training_data=1200, device=0, in_flight=1).predict_cell_r0_op(save=False).df
    auto = BattGP_Full(bd, max_training_data=1200, device=0)
    assert auto._in_flight([auto.packmodel, *auto.cellmodels], 300) == 9
    conc = auto.predict_cell_r0_op(save=False).df
    assert conc.shape == (300, 19) and conc.equals(seq)
    three = BattGP_Full(bd, max_training_data=1200, device=0, in_flight=3).predict_cell_r0_op(save=False).df
    assert three.equals(seq)


def test_system_driver_add_time_steps_large_m():
    """add_time_steps=True (battgp_full.py:86-96): M = N + age query points ride through the factorisation."""
    from battgp_amd.battgp_full import BattGP_Full
    from battgp_amd.synthetic import SyntheticBattData

    bd = SyntheticBattData("sys1", n_cells=1, age_days=400.0)
    sysm = BattGP_Full(bd, max_training_data=300, device=0)
    res = sysm.predict_cell_r0_op(add_time_steps=True, save=False)
    # t = 0 is both a training time and the start of the day grid: pandas' merge on "t" duplicates it,
    # exactly as in the reference (battgp_full.py:86-96,112)
    assert res.df["t"].nunique() == 300 + 400 - 1 and np.all(np.diff(res.df["t"]) >= 0)
    assert np.isfinite(res.df.to_numpy()).all()


def test_handle_pool_revives_handles_like_new_ones():
    """bgp_destroy parks the handle (streams, events, buffers), bgp_create revives it: it must behave like a
    fresh handle - default options, no kernel state, no stale data - and a same-sized problem reuses the buffers"""
    from battgp_amd.engine import EngineError, ExactGPEngine, trim_pool

    trim_pool()
    x, y = synthetic.make_cell_data(3000, seed=1)
    xq = synthetic.make_query(x, 50)
    a = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    a.set_options(nb_outer=256, lookahead=0)
    a.set_layout(512)
    lml_a, m_a, v_a = a.fit_predict(x, y, xq)
    a.close()
    b = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)  # revived
    with pytest.raises(EngineError):
        b.refit(synthetic.HYP_BATTGP)  # no data in this life of the handle
    with pytest.raises(EngineError):
        b.predict(xq)
    lml_b, m_b, v_b = b.fit_predict(x, y, xq)
    assert b.layout()[0] == 0  # default (automatic) layout again, not the forced slabs of the previous owner
    assert abs(lml_b - lml_a) <= 1e-9 * abs(lml_a) and np.allclose(m_b, m_a, rtol=1e-8, atol=0)
    bytes_b = b.device_bytes()
    b.close()
    c = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    x2, y2 = synthetic.make_cell_data(3000, seed=2)
    lml_c, m_c, _ = c.fit_predict(x2, y2, xq)
    assert c.device_bytes() == bytes_b  # same size: buffers found in place
    d = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)  # a second live handle is independent
    lml_d, m_d, _ = d.fit_predict(x2, y2, xq)
    assert lml_d == lml_c and np.array_equal(m_d, m_c)
    c.close()
    d.close()
    trim_pool()
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    lml_e, m_e, _ = e.fit_predict(x2, y2, xq)
    e.close()
    assert lml_e == lml_c and np.array_equal(m_e, m_c)


def test_handle_pool_is_thread_safe():
    """the reference trains cells from a thread pool (src/batt_models/battgp.py:191-216): handles are created and
    destroyed concurrently; parked handles must never be shared and results must not depend on the interleaving"""
    from battgp_amd.engine import ExactGPEngine, trim_pool

    trim_pool()
    cases = []
    for i, n in enumerate((300, 700, 700, 1100)):
        x, y = synthetic.make_cell_data(n, seed=50 + i)
        xq = synthetic.make_query(x, 40)
        e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
        lml, m, v = e.fit_predict(x, y, xq)
        e.close()
        cases.append((x, y, xq, lml, m, v))
    errors = []

    def worker(tid):
        try:
            for it in range(12):
                x, y, xq, lml, m, v = cases[(tid + it) % len(cases)]
                e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
                got = e.fit_predict(x, y, xq)
                e.close()
                assert got[0] == lml and np.array_equal(got[1], m) and np.array_equal(got[2], v)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    trim_pool()


# ---------------------------------------------------------------------------------------------
# round 2: trainers against torch autograd, residency of the training data, n_devices, Wiener wrapper
# ---------------------------------------------------------------------------------------------
def _torch_autograd_model(x, y, hyp0, ranges):
    """The reference's model restated in torch fp64 WITH autograd (CPU): raw parameters under the same constraints
    (sigmoid on the three Interval ranges, softplus on the lengthscales - cell_gp.py:182-194 quirk), Wiener + ARD-RBF
    kernel by direct differences, loss = -log_prob / N through torch.linalg.cholesky.  Checker only."""
    xt, yt = torch.as_tensor(x), torch.as_tensor(y)
    n = len(y)

    def inv_sig(v, lo, hi):
        p = (v - lo) / (hi - lo)
        return np.log(p) - np.log1p(-p)

    raw0 = np.concatenate(([inv_sig(hyp0[i], *ranges[i]) for i in range(3)], [v + np.log(-np.expm1(-v)) for v in hyp0[3:]]))
    raw = torch.tensor(raw0, dtype=torch.float64, requires_grad=True)
    lo = torch.tensor([r[0] for r in ranges], dtype=torch.float64)
    hi = torch.tensor([r[1] for r in ranges], dtype=torch.float64)

    def loss_fn():
        head = lo + (hi - lo) * torch.sigmoid(raw[:3])
        ls = torch.nn.functional.softplus(raw[3:])
        t = xt[:, :1]
        mn = torch.minimum(t, t.T)
        wien = mn**3 / 3 + (t - t.T).abs() * mn**2 / 2
        z = xt[:, 1:] / ls
        sq = ((z[:, None, :] - z[None, :, :]) ** 2).sum(-1)
        sigma = head[1] * wien + head[2] * torch.exp(-0.5 * sq) + head[0] * torch.eye(n, dtype=torch.float64)
        chol = torch.linalg.cholesky(sigma)
        zz = torch.linalg.solve_triangular(chol, yt.reshape(-1, 1), upper=False)
        lml = -0.5 * (zz**2).sum() - torch.log(torch.diagonal(chol)).sum() - 0.5 * n * np.log(2 * np.pi)
        return -lml / n

    return raw, loss_fn


def test_lbfgs_trainer_follows_torch_lbfgs_on_an_autograd_restatement():
    """the L-BFGS trainer must be torch.optim.LBFGS semantics (20 inner iterations per step, persistent
    history, lr).  Same optimiser on (a) the engine's loss/gradient and (b) a torch-autograd restatement of the
    reference model: identical loss trajectories (to the accuracy the two gradients agree)."""
    n = 160
    x, y = synthetic.make_cell_data(n, seed=21)
    hyp0 = np.array([4e-6, 1e-12, 0.02, 20.0, 20.0, 20.0])
    ranges = [cfg.NOISE_VARIANCE_RANGE, cfg.OUTPUTSCALE_WIENER_RANGE, cfg.OUTPUTSCALE_RBF_RANGE]
    cell = BatteryCellGP_Full(
        x, y, cellnr=1, noise_variance=hyp0[0], outputscale_wiener=hyp0[1], outputscale_rbf=hyp0[2], lengthscale_rbf=tuple(hyp0[3:])
    )
    max_iter, lr = 4, 0.5
    got = training.train_exact_gp_lbfgs(cell.model, max_iter=max_iter, rel_ftol=0.0, loss_scale=n, lr=lr, messages=False)

    raw, loss_fn = _torch_autograd_model(x, y, hyp0, ranges)
    opt = torch.optim.LBFGS([raw], line_search_fn="strong_wolfe", lr=lr)

    def closure():
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        return loss

    want = np.zeros(max_iter + 1) * np.nan
    for i in range(max_iter):
        with torch.no_grad():
            last = float(loss_fn()) * n
        want[i] = last
        opt.step(closure)
    want[-1] = last
    assert got.shape == want.shape
    assert np.allclose(got[:2], want[:2], rtol=1e-9)  # start point and the first full step (up to 20 inner iterations)
    assert np.allclose(got, want, rtol=1e-5), (got, want)
    assert got[max_iter - 1] < got[0] - 1.0  # it actually optimises
    assert np.allclose(cell.model.raw_vector(), raw.detach().numpy(), rtol=1e-3, atol=1e-3)
    assert np.all(np.isfinite(cell.model.raw_vector()))  # also for a parameter that saturated at its bound
    del cell.model


def test_adam_trainer_follows_torch_adam_on_an_autograd_restatement():
    n = 120
    x, y = synthetic.make_cell_data(n, seed=22)
    hyp0 = np.array([4e-6, 1e-12, 0.02, 20.0, 20.0, 20.0])
    ranges = [cfg.NOISE_VARIANCE_RANGE, cfg.OUTPUTSCALE_WIENER_RANGE, cfg.OUTPUTSCALE_RBF_RANGE]
    cell = BatteryCellGP_Full(
        x, y, cellnr=1, noise_variance=hyp0[0], outputscale_wiener=hyp0[1], outputscale_rbf=hyp0[2], lengthscale_rbf=tuple(hyp0[3:])
    )
    max_iter, lr = 12, 0.1
    got = training.train_exact_gp_adam(cell.model, max_iter=max_iter, rel_ftol=0.0, loss_scale=n, lr=lr, messages=False)
    raw, loss_fn = _torch_autograd_model(x, y, hyp0, ranges)
    opt = torch.optim.Adam([raw], lr=lr)
    want = np.zeros(max_iter + 1) * np.nan
    for i in range(max_iter):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        want[i] = float(loss) * n
        opt.step()
    want[-1] = want[max_iter - 1]  # the reference re-evaluates the LAST forward output (training.py:57)
    assert np.allclose(got, want, rtol=1e-7), (got, want)
    del cell.model


def test_training_data_replacement_reaches_the_gpu():
    """same-shaped replacement of train_inputs / train_targets must not reuse the stale copy in HBM."""
    x, y = synthetic.make_cell_data(300, seed=31)
    x2, y2 = synthetic.make_cell_data(300, seed=32)
    cell = BatteryCellGP_Full(x, y, cellnr=1)
    lml1 = cell.model.fit()
    cell.model.train_targets = torch.as_tensor(y2)
    cell.model.train_inputs = (torch.as_tensor(x2),)
    lml2 = cell.model.fit()
    want = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x2, y2).fit().lml
    assert abs(lml2 - want) < 1e-6 * abs(want) and abs(lml1 - lml2) > 1e-3
    # a hyper-parameter change alone takes the resident path (no upload) and still sees the new data
    cell.model.outputscale_rbf = 0.02
    hyp = synthetic.HYP_BATTGP.copy()
    hyp[2] = 0.02
    want = OracleGP(K.KERNEL_BATTGP, hyp, x2, y2).fit().lml
    assert abs(cell.model.fit() - want) < 1e-6 * abs(want)
    del cell.model


def test_n_devices_needs_a_process_group_of_that_size():
    from battgp_amd.engine import EngineError

    x, y = synthetic.make_cell_data(64, seed=1)
    with pytest.raises(EngineError, match="torch.distributed.run"):
        BatteryCellGP_Full(x, y, cellnr=1, n_devices=2)
    with pytest.raises(ValueError):
        BatteryCellGP_Full(x, y, cellnr=1, n_devices=0)


def test_hyperparameter_csv_layout(tmp_path):
    import pandas as pd

    x, y = synthetic.make_cell_data(80, seed=2)
    cell = BatteryCellGP_Full(x, y, cellnr=7)
    cell.marginallikelihood = -123.5
    cell.save_hyperparameters(str(tmp_path))
    df = pd.read_csv(tmp_path / "7hyperparams.csv", index_col=0)
    assert list(df.index) == [
        "Noise Variance", "Wiener Outputscale", "RBF Outputscale", "RBF Lengthscale 1", "RBF Lengthscale 2", "RBF Lengthscale 3",
        "Marginal Likelihood",
    ]
    assert list(df.columns) == ["params"]
    assert float(df.loc["RBF Lengthscale 2", "params"]) == cfg.LENGTHSCALE_RBF[1]
    assert float(df.loc["Marginal Likelihood", "params"]) == -123.5
    del cell.model


def test_wiener_rbf_kernel_wrapper_matches_oracle():
    """battgp_amd.wiener_kernel.WienerRBFKernel = `kernel(x1, x2).to_dense()` of the reference's composition
    (src/gp/wiener_kernel.py:10-32 inside src/batt_models/cell_gp.py:32-36)."""
    from battgp_amd.wiener_kernel import WienerRBFKernel

    x1, _ = synthetic.make_cell_data(90, seed=3)
    x2, _ = synthetic.make_cell_data(41, seed=4)
    k = WienerRBFKernel(synthetic.OUTPUTSCALE_WIENER, synthetic.OUTPUTSCALE_RBF, synthetic.LENGTHSCALE_RBF)
    hyp = np.concatenate(([0.0], synthetic.HYP_BATTGP[1:]))
    assert np.allclose(k(x1, x2), K.kernel_matrix(K.KERNEL_BATTGP, hyp, x1, x2), rtol=2e-13, atol=1e-300)
    sym = k(x1)
    assert np.allclose(sym, K.kernel_matrix(K.KERNEL_BATTGP, hyp, x1), rtol=2e-13, atol=1e-300)
    assert np.allclose(np.diag(sym), K.kernel_diag(K.KERNEL_BATTGP, hyp, x1), rtol=1e-13)  # diag=True branch
    k.close()
