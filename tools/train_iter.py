"""Cost of one optimiser iteration (LML value + analytic gradient) and of one fit+predict at the sizes the
reference actually runs (src/config.py:29: NB_DATAPOINTS = 1000, comment 16000).
    python tools/train_iter.py [N ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd import KERNEL_BATTGP, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [1000, 4000, 16000]:
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, 300)
    eng = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP, device=0)
    if os.environ.get("BGP_PANEL_SCHEME"):
        eng.set_panel_scheme(int(os.environ["BGP_PANEL_SCHEME"]))
    eng.fit(x, y)
    eng.lml_grad()
    reps = 20 if n <= 4000 else 5
    t0 = time.perf_counter()
    for i in range(reps):
        eng.refit(synthetic.HYP_BATTGP * (1.0 + 1e-3 * (i % 3)))
    t_refit = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for i in range(reps):
        eng.refit(synthetic.HYP_BATTGP * (1.0 + 1e-3 * (i % 3)))
        eng.lml_grad()
    t_iter = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for i in range(reps):
        eng.fit_predict(x, y, xq)
    t_fp = (time.perf_counter() - t0) / reps
    print(json.dumps({"n": n, "refit_ms": t_refit * 1e3, "refit_plus_grad_ms": t_iter * 1e3, "fit_predict_host_ms": t_fp * 1e3}), flush=True)
    eng.close()
