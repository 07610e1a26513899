"""TEST INFRASTRUCTURE ONLY: randomised runs of the MULTI-RANK sharded driver with its DeviceBackend on the CPU build of the
kernel sources (gloo between the ranks, every rank's streams deferred under a randomly chosen adversarial scheduler):
world size 2..4, ragged N, panel widths 64 / 128 - factorisation, prediction pass and the distributed in-place inverse of
the analytic gradient against the oracle, identical on every rank.  The CPU suite runs two fixed cases of this
(tests/test_emu_kernels.py::test_sharded_device_backend_ranks_on_the_cpu_build); this is the long form.

    python tests/emu/ranks_campaign.py [runs] [seed]"""
import os
import random
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    import torch.multiprocessing as mp

    import test_gpu_sharded as G  # the spawned ranks unpickle the worker by module name

    from battgp_amd import synthetic
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int.from_bytes(os.urandom(4), "little")
    rng = random.Random(seed)
    print(f"seed {seed}, {runs} runs", flush=True)
    os.environ["BGP_TEST_EMU"] = "1"
    ctx = mp.get_context("spawn")
    for run in range(runs):
        world = rng.choice([2, 2, 3, 4])
        nb = rng.choice([64, 128])
        n = rng.randint(nb + 1, 700)
        sched = rng.choice(["lazy", "eager", f"random:{rng.randint(0, 10**6)}", f"prio:{rng.randint(0, 23)}"])
        os.environ["HIPEMU_SCHED"] = sched
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=G._two_rank_worker, args=(r, world, port, n, nb, q)) for r in range(world)]
        [p.start() for p in procs]
        try:
            res = sorted(q.get(timeout=900) for _ in range(world))
        finally:
            for p in procs:
                p.join(120)
        assert all(p.exitcode == 0 for p in procs), (world, n, nb, sched, [p.exitcode for p in procs])
        x, y = synthetic.make_cell_data(n, seed=9)
        xq = synthetic.make_query(x, 33)
        ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
        m_ref, v_ref = ref.predict(xq)
        _, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
        for rank, lml, mean, var, grad in res:
            assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml), (world, n, nb, sched, rank)
            assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref), (world, n, nb, sched, rank)
            assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF, (world, n, nb, sched, rank)
            assert np.allclose(np.array(grad), g_ref, rtol=1e-5), (world, n, nb, sched, rank, grad, g_ref)
        assert all(r[1:] == res[0][1:] for r in res), (world, n, nb, sched)
        print(f"  run {run}: world {world}, N {n}, nb {nb}, {sched}: ok", flush=True)
    print(f"ok: {runs} multi-rank runs agree with the oracle", flush=True)


if __name__ == "__main__":
    main()
