"""Fresh-process latencies: import, bgp_create (HIP runtime start-up), first fit+predict (code-object load) vs warm.
    python tools/first_call.py"""
import sys, time
sys.path.insert(0, '/root/repo')
t0 = time.perf_counter()
from battgp_amd import KERNEL_BATTGP, synthetic
from battgp_amd.engine import ExactGPEngine
t1 = time.perf_counter()
x, y = synthetic.make_cell_data(1000); xq = synthetic.make_query(x, 300)
e = ExactGPEngine(KERNEL_BATTGP, synthetic.HYP_BATTGP)
t2 = time.perf_counter()
e.fit_predict(x, y, xq); t3 = time.perf_counter()
e.fit_predict(x, y, xq); t4 = time.perf_counter()
e.lml_grad(); t5 = time.perf_counter()
e.lml_grad(); t6 = time.perf_counter()
print(f"import {1e3*(t1-t0):.0f} ms, create {1e3*(t2-t1):.0f} ms, first fit_predict {1e3*(t3-t2):.1f} ms, second {1e3*(t4-t3):.2f} ms, first grad {1e3*(t5-t4):.1f}, second {1e3*(t6-t5):.2f}")
