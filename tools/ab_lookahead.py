"""A/B of the optional schedules of the blocked Cholesky (lookahead word: +32 slim chain kernels, +64 split panels, +128 update + next tile Cholesky in one launch)
against the default, at the sizes where the panel chain matters.  One JSON line per (N, lookahead); bench.py runs this
in a child process with a time limit and files the lines under "experiments" (a schedule that has never met the GPU
must not be able to take the headline measurement down with it).

    python tools/ab_lookahead.py [reps] [N ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from battgp_amd import KERNEL_BATTGP, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sizes = [int(a) for a in sys.argv[2:]] or [16384, 40000]
m = 300
for n in sizes:
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, m)
    tx, ty, tq = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, xq))
    tm = torch.empty(m, dtype=torch.float64, device="cuda")
    tv = torch.empty(m, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    base = None
    for la in (1, 1 | 32, 1 | 64, 1 | 32 | 64, 1 | 128, 1 | 64 | 128):
        eng = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP, device=0)
        eng.set_options(lookahead=la)
        eng.set_panel_scheme(1)
        ts = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            eng.fit_predict_device(tx.data_ptr(), ty.data_ptr(), n, 4, tq.data_ptr(), m, tm.data_ptr(), tv.data_ptr())
            ts.append(time.perf_counter() - t0)
        mean = tm.cpu().numpy()
        sig = (eng.lml, float(mean[0]), float(mean[-1]))
        base = base or sig
        print(json.dumps({"n": n, "lookahead": la, "fit_predict_ms": float(np.median(ts[1:])) * 1e3, "potrf_ms": eng.phase_times()["potrf_ms"],
                          "identical_to_default": sig == base}), flush=True)
        eng.close()
