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
    ph = eng.phase_times()  # device-side split of the last iteration: the fit's phases and the gradient pass
    # a prediction at the point just differentiated: the factor comes back by a re-run of the fit (default) ...
    t0 = time.perf_counter()
    eng.predict(xq)
    t_pred_refit = time.perf_counter() - t0
    restore_refit = eng.phase_times()["restore_ms"]
    # ... or by one device-to-device copy (bgp_set_keep_factor)
    eng.set_keep_factor(True)
    eng.lml_grad()
    t0 = time.perf_counter()
    eng.predict(xq)
    t_pred_keep = time.perf_counter() - t0
    restore_keep = eng.phase_times()["restore_ms"]
    eng.set_keep_factor(False)
    t0 = time.perf_counter()
    for i in range(reps):
        eng.fit_predict(x, y, xq)
    t_fp = (time.perf_counter() - t0) / reps
    print(json.dumps({"n": n, "refit_ms": t_refit * 1e3, "refit_plus_grad_ms": t_iter * 1e3, "fit_predict_host_ms": t_fp * 1e3,
                      "grad_over_refit": (t_iter - t_refit) / t_refit, "device_ms": {k: ph[k] for k in ("fill_ms", "potrf_ms", "solve_ms", "grad_ms")},
                      "predict_after_grad_ms": {"refit": t_pred_refit * 1e3, "restore_ms_refit": restore_refit,
                                                "keep_factor": t_pred_keep * 1e3, "restore_ms_keep_factor": restore_keep}}), flush=True)
    eng.close()
