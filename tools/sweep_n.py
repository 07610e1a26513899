"""fit+predict wall-clock and GFLOP/s vs N on one MI355X (BASELINE.json's metric is quoted "vs N").
Warm handle, median of `reps` fused fit+predict calls with the inputs resident in HBM.

    python tools/sweep_n.py [reps] [N ...]        -> one JSON line per (kernel, N)
Environment (experiments): BGP_NB = outer panel width, BGP_SCHEME = panel scheme 0 / 1, BGP_LA = look-ahead word,
BGP_ONLY = battgp | scaled_rbf.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from battgp_amd import KERNEL_BATTGP, KERNEL_SCALED_RBF, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sizes = [int(a) for a in sys.argv[2:]] or [2048, 4096, 8192, 16384, 32768, 40000, 65536]
m = 300
for kname, kid in (("battgp", KERNEL_BATTGP), ("scaled_rbf", KERNEL_SCALED_RBF)):
    for n in sizes:
        if os.environ.get("BGP_ONLY", kname) != kname:
            continue
        if kname == "scaled_rbf" and n not in (2048, 40000):  # BASELINE configs 1 and 2 name the plain RBF too
            continue
        x, y = synthetic.make_cell_data(n)
        xq = synthetic.make_query(x, m)
        hyp = synthetic.HYP_BATTGP
        if kname == "scaled_rbf":  # ScaledRBFModel works on standardised inputs (one isotropic lengthscale)
            mu, sd = x.mean(axis=0), x.std(axis=0)
            x, xq, y = (x - mu) / sd, (xq - mu) / sd, y - y.mean()
            hyp = np.array([2.33e-6, 0.0099, 1.5])
        tx, ty, tq = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, xq))
        tm = torch.empty(m, dtype=torch.float64, device="cuda")
        tv = torch.empty(m, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        eng = ExactGPEngine(kid, hyp, device=0)
        if "BGP_NB" in os.environ or "BGP_LA" in os.environ:
            eng.set_options(nb_outer=int(os.environ.get("BGP_NB", -1)), lookahead=int(os.environ.get("BGP_LA", -1)))
        if "BGP_SCHEME" in os.environ:
            eng.set_panel_scheme(int(os.environ["BGP_SCHEME"]))
        ts = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            eng.fit_predict_device(tx.data_ptr(), ty.data_ptr(), n, 4, tq.data_ptr(), m, tm.data_ptr(), tv.data_ptr())
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[1:]))
        ph = eng.phase_times()
        res = eng.residuals(128)
        flop = n**3 / 3.0 + float(n) * n * m + 2.0 * n * n
        print(json.dumps({
            "kernel": kname, "n": n, "m": m, "nb": os.environ.get("BGP_NB"), "scheme": os.environ.get("BGP_SCHEME"), "la": os.environ.get("BGP_LA"), "fit_predict_ms": t * 1e3, "gflops": flop / t / 1e9,
            "potrf_ms": ph["potrf_ms"], "fill_ms": ph["fill_ms"], "fill_gbs": ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9,
            "lml": eng.lml, "jitter": eng.jitter, "rel_solve": res[0], "max_llt": res[1],
        }), flush=True)
        eng.close()
