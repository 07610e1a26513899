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
    e.close()
    gp.engine.close()


def test_sharded_matern_and_jitter():
    x, y = synthetic.make_cell_data(900, seed=1)
    gp = make_sharded_gp(K.KERNEL_MATERN32, synthetic.HYP_MATERN32, nb=256)
    lml = gp.fit(x, y)
    assert abs(lml - OracleGP(K.KERNEL_MATERN32, synthetic.HYP_MATERN32, x, y).fit().lml) < 1e-6 * abs(lml)
    gp.engine.close()
    # exactly singular matrix -> first jitter rung, same as the single-GPU engine
    xs, ys = np.zeros((130, 2)), np.ones(130)
    gp = make_sharded_gp(K.KERNEL_BATTGP, np.array([0.0, 1.0, 1.0, 1.0]), nb=64)
    gp.fit(xs, ys)
    assert gp.jitter == 1e-8
    gp.engine.close()


def _two_rank_worker(rank, world, port, n, nb, q):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    # both ranks share the one GPU of the test box; gloo moves the device buffers (RCCL refuses two ranks
    # on one device) - the schedule, packing and offsets are exactly those of the multi-GPU run
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from battgp_amd import parallel, synthetic
    from battgp_amd.sharded import make_sharded_gp

    x, y = synthetic.make_cell_data(n, seed=9)
    xq = synthetic.make_query(x, 33)
    gp = make_sharded_gp(0, synthetic.HYP_BATTGP, nb=nb, backend_name="gloo")
    lml = gp.fit(x, y)
    mean, var = gp.predict(xq)
    q.put((rank, lml, mean.tolist(), var.tolist()))
    parallel.barrier(gp.dist)
    gp.engine.close()
    gp.dist.destroy_process_group()


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
    for rank, lml, mean, var in res:
        assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
        assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref)
        assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
    assert res[0][1:] == res[1][1:]
