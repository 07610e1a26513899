"""TEST INFRASTRUCTURE ONLY: checks the stream / event graph of the blocked Cholesky's schedules on the CPU build.

The CPU build's stream model (tests/emu/hipemu.cpp, HIPEMU_SCHED) can defer every launch and run the queues in an order
that honours ONLY in-stream order, hipStreamWaitEvent edges and host synchronisation - with each of the four stream
roles (main, panel, copy, bulk) in turn allowed to run as far ahead as the graph permits.  Because every schedule of
the engine is bit-identical by construction, a missing edge shows up as a result that differs from the immediate-mode
run (or as a failed factorisation).

    python tests/emu/stream_graph_check.py            -> all 24 priority permutations + lazy / eager / random, every
                                                          look-ahead word; one line per policy
    python tests/emu/stream_graph_check.py --mutate   -> additionally drops every hipStreamWaitEvent of the split
                                                          schedule in turn and reports which policies notice
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
from inject import installed  # noqa: E402

LOOKAHEADS = (0, 1, 2, 1 | 8, 1 | 32, 1 | 64, 1 | 32 | 64, 1 | 128, 1 | 64 | 128)
POLICIES = ["lazy", "eager", "random:1", "random:2"] + [f"prio:{p}" for p in range(24)]


def workload(la, n=320, nb=64, m=24, kid=0):
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine

    hyp = synthetic.HYP_BATTGP if kid == 0 else synthetic.HYP_MATERN32
    x, y = synthetic.make_cell_data(n, seed=9)
    xq = synthetic.make_query(x, m)
    e = ExactGPEngine(kid, hyp)
    try:
        e.set_options(nb_outer=nb, lookahead=la)
        e.set_panel_scheme(1)
        lml, mean, var = e.fit_predict(x, y, xq, min_var=-1.0)
        m2, v2 = e.predict(xq[:7], min_var=-1.0)
        g = e.lml_grad()
        return (np.float64(lml), mean, var, m2, v2, g, e.factor_diag())
    finally:
        e.close()


def same(a, b):
    return all(np.array_equal(u, v) for u, v in zip(a, b))


def run_policy(lib, policy, las, ref):
    lib.hipemu_set_sched(policy.encode())
    bad = []
    for la in las:
        try:
            out = workload(la)
            if not same(out, ref):
                bad.append(la)
        except Exception:  # a failed factorisation is a detection too
            bad.append(la)
    lib.hipemu_set_sched(b"sync")
    return bad


def main(argv):
    with installed(experimental=True) as lib:  # the split-panel schedule (bulk stream) is part of the sweep
        lib.hipemu_set_sched.argtypes = [C.c_char_p]
        lib.hipemu_drop_wait.argtypes = [C.c_long]
        lib.hipemu_wait_count.restype = C.c_long
        lib.hipemu_set_sched(b"sync")
        ref = workload(1)
        assert all(same(workload(la), ref) for la in LOOKAHEADS)  # immediate mode: the schedules agree bit for bit
        report = {"policies": {}, "mutations": []}
        for pol in POLICIES:
            bad = run_policy(lib, pol, LOOKAHEADS, ref)
            report["policies"][pol] = bad
            print(f"{pol:10s} {'ok' if not bad else 'DIFFERS for lookahead ' + str(bad)}", flush=True)
        if "--mutate" in argv:
            la = 1 | 32 | 64
            lib.hipemu_set_sched(b"lazy")
            lib.hipemu_drop_wait(-1)
            workload(la)
            nwaits = lib.hipemu_wait_count()
            lib.hipemu_set_sched(b"sync")
            pols = [f"prio:{p}" for p in range(24)]
            for k in range(nwaits):
                caught = []
                for pol in pols:
                    lib.hipemu_set_sched(pol.encode())
                    lib.hipemu_drop_wait(k)
                    try:
                        ok = same(workload(la), ref)
                    except Exception:
                        ok = False
                    lib.hipemu_drop_wait(-1)
                    lib.hipemu_set_sched(b"sync")
                    if not ok:
                        caught.append(pol)
                report["mutations"].append({"wait": k, "caught_by": len(caught)})
                print(f"drop wait {k:3d} of {nwaits}: noticed by {len(caught):2d} of {len(pols)} priority orders", flush=True)
        ok = not any(report["policies"].values())
        print(json.dumps({"ok": ok, "n_policies": len(POLICIES), "lookaheads": list(LOOKAHEADS),
                          "mutations_total": len(report["mutations"]),
                          "mutations_noticed": sum(1 for mrec in report["mutations"] if mrec["caught_by"] > 0)}))
        return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
