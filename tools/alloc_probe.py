"""Where does the per-handle overhead go?  (the reference builds one model object per cell:
src/batt_models/battgp_full.py:41-60, so bgp_create / first fit / bgp_destroy are paid per cell)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd import KERNEL_BATTGP, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [1000, 4000, 16000, 40000]:
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, 300)
    warm = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP)
    warm.fit_predict(x, y, xq)
    t0 = time.perf_counter()
    warm.fit_predict(x, y, xq)
    reuse = time.perf_counter() - t0
    warm.close()
    acc = {"create": 0.0, "first_fit": 0.0, "second_fit": 0.0, "destroy": 0.0}
    reps = 5
    for _ in range(reps):
        t0 = time.perf_counter()
        e = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP)
        t1 = time.perf_counter()
        e.fit_predict(x, y, xq)
        t2 = time.perf_counter()
        e.fit_predict(x, y, xq)
        t3 = time.perf_counter()
        e.close()
        t4 = time.perf_counter()
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += v / reps
    print(json.dumps({"n": n, "warm_fit_predict_ms": reuse * 1e3, **{k + "_ms": v * 1e3 for k, v in acc.items()}}), flush=True)
