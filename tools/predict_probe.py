"""Cost of a LATER prediction on a fitted model (the first one rides the factorisation): the reference's plotting code
calls predict() on several grids of the same model (gp_operational_char_figure.py).   python tools/predict_probe.py [N ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd import KERNEL_BATTGP, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4000, 16000, 40000]:
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, 300)
    eng = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP, device=0)
    lml, m0, v0 = eng.fit_predict(x, y, xq)
    m1, v1 = eng.predict(xq)  # builds whatever the later passes cache
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        m1, v1 = eng.predict(xq)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        mm = eng.predict(xq, want_var=False)
    dtm = (time.perf_counter() - t0) / reps
    import numpy as np
    print(json.dumps({"n": n, "later_predict_ms": dt * 1e3, "mean_only_ms": dtm * 1e3,
                      "rel_mean_vs_fused": float(np.linalg.norm(m1 - m0) / np.linalg.norm(m0)),
                      "max_var_diff": float(np.max(np.abs(v1 - v0)))}), flush=True)
    eng.close()
