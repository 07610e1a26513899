"""Sharded-Cholesky driver on the GPU with its DeviceBackend (HIP kernels through the C-ABI).
A 1-GPU box can only run world = 1, which still exercises every device entry point the multi-rank
path uses (panel fill, panel factorisation, packed panel, per-panel SYRK, prediction pass); the
multi-rank schedule itself is covered by tests/test_sharded_cpu.py over gloo."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402
from battgp_amd.sharded import make_sharded_gp  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402


@pytest.mark.parametrize("n,nb", [(700, 128), (1500, 256), (2100, 512)])
def test_sharded_single_rank_matches_oracle_and_engine(n, nb):
    x, y = synthetic.make_cell_data(n, seed=n)
    xq = synthetic.make_query(x, 57)
    gp = make_sharded_gp(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, nb=nb)
    lml = gp.fit(x, y)
    mean, var = gp.predict(xq)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    assert gp.jitter == 0.0
    assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
    assert np.linalg.norm(mean - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    assert np.max(np.abs(var - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    assert abs(e.fit(x, y) - lml) < 1e-9 * abs(lml)
    g_eng = e.lml_grad()
    e.close()
    # analytic LML gradient: Sigma^-1 in place over the panels (consumes the factor), then predictions as before
    from oracle.exact_gp import lml_and_grad

    _, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    grad = gp.lml_grad()
    assert np.allclose(grad, g_ref, rtol=1e-5), (grad, g_ref)
    assert np.allclose(grad, g_eng, rtol=1e-6), (grad, g_eng)
    mean_b, var_b = gp.predict(xq)
    assert np.array_equal(mean_b, mean) and np.array_equal(var_b, var)
    # fit + first prediction in ONE pass: the query rows ride through the panels (bgp_cross_block_dev); same LML bit
    # for bit (the riding rows do not touch the matrix), posterior within the parity tolerance (other summation order);
    # later queries and the gradient work on the store that carries the riding rows
    lml_f, mean_f, var_f = gp.fit_predict(x, y, xq)
    assert lml_f == lml
    assert np.linalg.norm(mean_f - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    assert np.max(np.abs(var_f - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    mean_c, var_c = gp.predict(xq[:20])
    assert np.array_equal(mean_c, mean[:20]) and np.array_equal(var_c, var[:20])
    assert np.allclose(gp.lml_grad(), grad, rtol=1e-12)
    # a second fit on the same object (resident buffers, new hyper-parameters) and the timers
    hyp2 = synthetic.HYP_BATTGP.copy()
    hyp2[2] *= 2.0
    gp.set_hyp(hyp2)
    with pytest.raises(RuntimeError, match="fit first"):
        gp.predict(xq)
    lml2 = gp.fit(x, y)
    assert abs(lml2 - OracleGP(K.KERNEL_BATTGP, hyp2, x, y).fit().lml) < 1e-6 * abs(lml2)
    assert set(gp.timers()) == {"fit_s", "predict_s", "grad_s", "fit_predict_s"}
    gp.close()


def test_sharded_matern_and_jitter():
    x, y = synthetic.make_cell_data(900, seed=1)
    gp = make_sharded_gp(K.KERNEL_MATERN32, synthetic.HYP_MATERN32, nb=256)
    lml = gp.fit(x, y)
    assert abs(lml - OracleGP(K.KERNEL_MATERN32, synthetic.HYP_MATERN32, x, y).fit().lml) < 1e-6 * abs(lml)
    gp.close()
    # exactly singular matrix -> first jitter rung, same as the single-GPU engine
    xs, ys = np.zeros((130, 2)), np.ones(130)
    gp = make_sharded_gp(K.KERNEL_BATTGP, np.array([0.0, 1.0, 1.0, 1.0]), nb=64)
    gp.fit(xs, ys)
    assert gp.jitter == 1e-8
    gp.close()
    # failure in a LATER panel: the device flag poisons the enqueued pipeline, the host sees it once at the end
    xs[:64, 1] = 3.0 * np.arange(64)
    gp = make_sharded_gp(K.KERNEL_BATTGP, np.array([0.0, 1.0, 1.0, 1.0]), nb=64)
    gp.fit(xs, ys)
    assert gp.jitter == 1e-8
    from battgp_amd.engine import NotPSDError

    gp.max_tries = 0  # the plain attempt only
    with pytest.raises(NotPSDError, match="leading minor 65"):
        gp.fit(xs, ys)
    gp.close()


def _emu_hook(root):
    """tests/test_emu_kernels.py / `pytest --emu` with BGP_TEST_EMU=1: the spawned ranks run on the CPU build of the kernel
    sources (test infrastructure; the "device" buffers are host memory, the streams follow HIPEMU_SCHED)"""
    import os
    import sys

    if os.environ.get("BGP_TEST_EMU") == "1":
        sys.path.insert(0, os.path.join(root, "tests", "emu"))
        from inject import fake_cuda_tensors, installed

        fake_cuda_tensors()
        installed().__enter__()


def _two_rank_worker(rank, world, port, n, nb, q):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    # both ranks share the one GPU of the test box; gloo moves the device buffers (RCCL refuses two ranks
    # on one device) - the schedule, packing and offsets are exactly those of the multi-GPU run
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _emu_hook(root)
    from battgp_amd import parallel, synthetic
    from battgp_amd.sharded import make_sharded_gp

    x, y = synthetic.make_cell_data(n, seed=9)
    xq = synthetic.make_query(x, 33)
    gp = make_sharded_gp(0, synthetic.HYP_BATTGP, nb=nb, backend_name="gloo")
    lml = gp.fit(x, y)
    mean, var = gp.predict(xq)
    grad = gp.lml_grad()
    lml_f, mean_f, var_f = gp.fit_predict(x, y, xq)  # the fused pass: riding rows, no right-looking prediction pass
    assert lml_f == lml
    assert np.linalg.norm(mean_f - mean) < 1e-8 * np.linalg.norm(mean) and np.max(np.abs(var_f - var)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    assert "reduce" not in gp.comm_bytes().get("fit", {}) and gp.comm_bytes()["predict"]["reduce"][0] == gp.lay.npanels
    q.put((rank, lml, mean.tolist(), var.tolist(), grad.tolist()))
    parallel.barrier(gp.dist)
    dist = gp.dist
    gp.close()
    dist.destroy_process_group()


@pytest.mark.timeout(280)
def test_sharded_two_ranks_device_backend_over_gloo():
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, nb, world = 1900, 256, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n, nb, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    x, y = synthetic.make_cell_data(n, seed=9)
    xq = synthetic.make_query(x, 33)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    from oracle.exact_gp import lml_and_grad

    _, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    for rank, lml, mean, var, grad in res:
        assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
        assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref)
        assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
        assert np.allclose(np.array(grad), g_ref, rtol=1e-5), (rank, grad, g_ref)
    assert res[0][1:] == res[1][1:]


def _rccl_world1_worker(port, n, nb, q):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BGP_FORCE_GROUP="1")
    import torch.distributed as dist

    from battgp_amd import synthetic
    from battgp_amd.sharded import make_sharded_gp

    stage = "rccl"
    try:
        # can RCCL form a communicator and move a byte on this box at all?  (an environment question, not the engine's)
        import torch

        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        probe = torch.ones(4, dtype=torch.float64, device="cuda")
        dist.all_reduce(probe)
        dist.broadcast(probe, src=0)
        torch.cuda.synchronize()
        assert float(probe.sum()) == 4.0
        stage = "engine"
        x, y = synthetic.make_cell_data(n, seed=11)
        xq = synthetic.make_query(x, 33)
        gp = make_sharded_gp(0, synthetic.HYP_BATTGP, nb=nb, backend_name="nccl")
        assert gp.dist is not None and dist.get_backend() == "nccl"
        lml = gp.fit(x, y)
        mean, var = gp.predict(xq)
        grad = gp.lml_grad()  # its broadcasts / all-reduces (async, waited on the device) go through RCCL as well
        q.put(("ok", lml, mean.tolist(), var.tolist(), grad.tolist()))
        gp.close()
        dist.destroy_process_group()
    except BaseException as exc:  # noqa: BLE001 - reported to the parent, which decides
        q.put((stage, f"{type(exc).__name__}: {exc}"))
        raise


@pytest.mark.timeout(280)
def test_sharded_one_rank_group_over_rccl():
    """A one-process "nccl" group: every broadcast / all-reduce / reduce of the sharded schedule goes through
    ProcessGroupNCCL (= RCCL) on the engine's stream (ExternalStream hand-off, async work objects waited on the
    device) - the calls a multi-GPU node issues, with one rank.  Result must equal the oracle."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, nb = 1900, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, n, nb, q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    if res[0] == "rccl":
        pytest.skip(f"RCCL cannot run a one-rank group on this box ({res[1]}): environment, not the engine")
    assert res[0] == "ok", res
    assert p.exitcode == 0
    _, lml, mean, var, grad = res
    x, y = synthetic.make_cell_data(n, seed=11)
    xq = synthetic.make_query(x, 33)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
    assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref)
    assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    from oracle.exact_gp import lml_and_grad

    assert np.allclose(np.array(grad), lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)[1], rtol=1e-5)


def _colmajor(a, ld=None):
    m, n = a.shape
    ld = m if ld is None else ld
    t = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    t[:, :m] = torch.from_numpy(np.ascontiguousarray(a.T))
    return t


def test_factor_pack_panel_dev_against_numpy():
    """bgp_factor_pack_panel_dev: diagonal-block chain + one TRSM GEMM straight into the packed buffer"""
    import ctypes as C

    from battgp_amd import _lib

    rng = np.random.default_rng(3)
    rows, nbk, ld = 2304, 512, 2400
    g = rng.normal(size=(nbk, nbk))
    top = g @ g.T + nbk * np.eye(nbk)
    a = np.vstack([top, rng.normal(size=(rows - nbk, nbk))])
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    e.set_options(nb_outer=512)
    lib = _lib.load()
    panel = _colmajor(a, ld)
    pack = torch.full((nbk * rows + 1,), float("nan"), dtype=torch.float64, device="cuda")
    inv = torch.empty((nbk // 64) * 4096, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    info = C.c_int(-7)
    rc = lib.bgp_factor_pack_panel_dev(e._h, C.c_void_p(panel.data_ptr()), ld, rows, nbk, C.c_void_p(inv.data_ptr()),
                                       C.c_void_p(pack.data_ptr()), C.byref(info))
    assert rc == 0 and info.value == 0
    l11 = np.linalg.cholesky(top)
    l21 = np.linalg.solve(l11, a[nbk:].T).T
    got_panel = panel[:, :rows].cpu().numpy().T
    got_pack = pack[: nbk * rows].view(nbk, rows).cpu().numpy().T
    assert np.allclose(np.tril(got_panel[:nbk]), l11, rtol=1e-11, atol=1e-11)
    assert np.allclose(got_panel[nbk:], l21, rtol=1e-10, atol=1e-11)
    assert np.allclose(np.tril(got_pack[:nbk]), l11, rtol=1e-11, atol=1e-11)
    assert np.array_equal(got_pack[nbk:], got_panel[nbk:])  # the packed copy IS what was written back
    # a non-positive pivot is reported relative to the panel
    bad = a.copy()
    bad[200, 200] = -1.0
    panel = _colmajor(bad, ld)
    torch.cuda.synchronize()
    rc = lib.bgp_factor_pack_panel_dev(e._h, C.c_void_p(panel.data_ptr()), ld, rows, nbk, C.c_void_p(inv.data_ptr()),
                                       C.c_void_p(pack.data_ptr()), C.byref(info))
    assert rc == 0 and info.value == 201
    e.close()


def test_update_panels_dev_against_numpy():
    """bgp_update_panels_dev: all rank-k updates of one step in one call, launches alternating between two streams"""
    import ctypes as C

    from battgp_amd import _lib

    rng = np.random.default_rng(4)
    k, rows_p = 256, 1664  # packed panel P [rows_p, k]
    p = rng.normal(size=(rows_p, k))
    ld = 1800
    # three local panels of 256 / 256 / 128 columns whose rows start at offsets 256, 768, 1280 of P
    panels = [(0, 256, 256), (256, 256, 768), (512, 128, 1280)]  # (local col0, width, offset in P)
    store = rng.normal(size=(ld, 640))
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    lib = _lib.load()
    ts, tp = _colmajor(store), _colmajor(p)
    torch.cuda.synchronize()
    desc, want = [], store.copy()
    for lc, w, off in panels:
        rows_j = rows_p - off
        r0 = 100 + off  # where this panel's diagonal sits in the store (any row offset works)
        desc.append([lc * ld + r0, rows_j, w, off, ld])
        upd = p[off:] @ p[off : off + w].T
        ti = np.arange(rows_j)[:, None] // 128
        tj = np.arange(w)[None, :] // 128
        blk = want[r0 : r0 + rows_j, lc : lc + w]
        blk[ti >= tj] -= upd[ti >= tj]
    d = np.ascontiguousarray(desc, dtype=np.int64)
    flag = torch.zeros(1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = lib.bgp_update_panels_dev(e._h, C.c_void_p(ts.data_ptr()), d.ctypes.data_as(C.POINTER(C.c_int64)), len(desc),
                                   C.c_void_p(tp.data_ptr()), rows_p, k, C.c_void_p(flag.data_ptr()))
    assert rc == 0
    assert lib.bgp_sync(e._h) == 0
    got = ts.cpu().numpy().T
    assert np.allclose(got, want, rtol=1e-12, atol=1e-10)
    # a set abort flag (int in the low word of the slot) turns the launches into no-ops
    flag.view(torch.int32)[0] = 7
    torch.cuda.synchronize()
    rc = lib.bgp_update_panels_dev(e._h, C.c_void_p(ts.data_ptr()), d.ctypes.data_as(C.POINTER(C.c_int64)), len(desc),
                                   C.c_void_p(tp.data_ptr()), rows_p, k, C.c_void_p(flag.data_ptr()))
    assert rc == 0 and lib.bgp_sync(e._h) == 0
    assert np.array_equal(ts.cpu().numpy().T, got)
    e.close()


def _plugin_worker(rank, world, port, n, q):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _emu_hook(root)
    import torch.distributed as dist

    from battgp_amd import synthetic
    from battgp_amd.battcellgp_full import BatteryCellGP_Full
    from battgp_amd.operating_point import Op

    dist.init_process_group("gloo")
    x, y = synthetic.make_cell_data(n, seed=12)
    cell = BatteryCellGP_Full(x, y, cellnr=3, n_devices=world, device=0)  # the reference's multi-GPU switch
    t = np.linspace(x[0, 0], x[-1, 0], 40)
    df = cell.predict_r0_op(Op(*synthetic.REF_OP), t)
    assert "fit_predict_s" in cell.model._shard().timers()  # first call, variance wanted, 40 queries: the fused pass
    # a mean-only first call and a query block that is large next to the matrix do NOT ride through the panels
    # (fit, then the right-looking pass over the stored factor) - same numbers
    xq40 = np.column_stack((t, *(np.full(40, v) for v in synthetic.REF_OP)))
    c2 = BatteryCellGP_Full(x, y, cellnr=5, n_devices=world, device=0)
    mean_only = c2.model.posterior_mean(xq40)
    assert "fit_predict_s" not in c2.model._shard().timers() and c2.model._shard().lay.ride == 0
    assert np.linalg.norm(mean_only - df["r0_acausal_c3"].to_numpy()) < 1e-9 * np.linalg.norm(mean_only)
    del c2.model
    tb = np.linspace(x[0, 0], x[-1, 0] + 30.0, 1100)  # M > max(1024, N / 8): battgp_full.py:86-96's add_time_steps shape
    xqb = np.column_stack((tb, *(np.full(1100, v) for v in synthetic.REF_OP)))
    c3 = BatteryCellGP_Full(x, y, cellnr=6, n_devices=world, device=0)
    mb, vb = c3.model.posterior(xqb)
    assert "fit_predict_s" not in c3.model._shard().timers() and c3.model._shard().lay.ride == 0
    mb_ref, vb_ref = cell.model.posterior(xqb)  # the fitted model's later prediction
    assert np.allclose(mb.numpy(), mb_ref.numpy(), rtol=1e-9, atol=0) and np.allclose(vb.numpy(), vb_ref.numpy(), rtol=1e-7, atol=1e-12 * synthetic.OUTPUTSCALE_RBF)
    del c3.model
    loss = cell.model.neg_mll() * n
    loss2, raw_grad = cell.model.neg_mll_and_raw_grad()  # analytic, sharded: what loss.backward() gives the reference
    assert loss2 * n == loss
    # three optimiser iterations through the plugin (the reference's train_hyperparameters, battcellgp_full.py:127-166):
    # every iteration = one sharded fit + one distributed in-place inverse
    trained = BatteryCellGP_Full(x, y, cellnr=4, n_devices=world, device=0, max_iter=3)
    losses = trained.train_hyperparameters(messages=False)
    q.put((rank, df["r0_acausal_c3"].tolist(), df["r0var_acausal_c3"].tolist(), loss, raw_grad.tolist(), np.asarray(losses).tolist()))
    del trained.model
    dist.barrier()
    del cell.model
    dist.destroy_process_group()


@pytest.mark.timeout(280)
def test_plugin_n_devices_routes_to_the_sharded_engine():
    """BatteryCellGP_Full(..., n_devices=2) inside a 2-rank process group = ONE GP over two ranks (here sharing the
    test box's single GPU over gloo): same posterior and loss as the oracle, identical on both ranks."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 1500, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plugin_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    x, y = synthetic.make_cell_data(n, seed=12)
    t = np.linspace(x[0, 0], x[-1, 0], 40)
    xq = np.column_stack((t, np.full(40, synthetic.REF_OP[0]), np.full(40, synthetic.REF_OP[1]), np.full(40, synthetic.REF_OP[2])))
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    # the same model on ONE device: loss and raw-parameter gradient of the single-GPU engine
    from battgp_amd.battcellgp_full import BatteryCellGP_Full

    one = BatteryCellGP_Full(x, y, cellnr=3, device=0, max_iter=3)
    _, raw_one = one.model.neg_mll_and_raw_grad()
    losses_one = np.asarray(one.train_hyperparameters(messages=False))
    del one.model
    for rank, mean, var, loss, raw_grad, losses in res:
        # the optimiser walks the same path on two ranks as on one device
        assert np.allclose(np.asarray(losses), losses_one, rtol=1e-7, equal_nan=True), (losses, losses_one)
        assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref)
        assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
        assert abs(loss - ref.neg_mll_scaled) < 1e-6 * abs(ref.neg_mll_scaled)
        assert np.allclose(np.array(raw_grad), raw_one, rtol=1e-6, atol=1e-12 * np.abs(raw_one).max()), (raw_grad, raw_one)
    assert res[0][1:] == res[1][1:]
