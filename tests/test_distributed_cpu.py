"""world_size-2 gloo tests (CPU) of the multi-process plumbing used by bench.py --gpus N and by the
one-cell-per-GPU deployment: barrier + max-over-ranks timing, per-rank cell assignment, result gather."""

import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from battgp_amd import parallel, synthetic
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    dist = parallel.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    cells = parallel.cells_for_rank([-1, 1, 2], rank, world)
    # every rank works on its OWN cell(s): stand-in for the per-GPU fit (the oracle replaces the GPU
    # engine here only because this test runs on CPU; the communication pattern is what is tested)
    means = []
    for c in cells:
        x, y = synthetic.make_cell_data(64, seed=100 + c)
        gp = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
        m, _ = gp.predict(synthetic.make_query(x, 8))
        means.append(m)
    parallel.barrier(dist)
    t = parallel.max_over_ranks(dist, 1.0 + rank)  # slowest rank defines the job time
    gathered = parallel.gather_vectors(dist, means[0])
    q.put((rank, cells, t, [g.tolist() for g in gathered]))
    parallel.barrier(dist)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_independent_cells_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, c0, t0, g0), (r1, c1, t1, g1) = results
    assert c0 == [-1, 2] and c1 == [1]
    assert t0 == t1 == 2.0  # max over ranks
    assert g0 == g1 and len(g0) == 2  # all-gather delivered both ranks' vectors everywhere
    assert not np.allclose(g0[0], g0[1])  # different cells -> different posteriors
