"""Timing of fit vs fit+gradient at a realistic training size (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd import synthetic
from battgp_amd.engine import ExactGPEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
x, y = synthetic.make_cell_data(n)
e = ExactGPEngine(0, synthetic.HYP_BATTGP)
e.fit(x, y)
for rep in range(2):
    t0 = time.perf_counter(); e.refit(synthetic.HYP_BATTGP); t1 = time.perf_counter(); g = e.lml_grad(); t2 = time.perf_counter()
    print(f"N={n}: refit {1e3*(t1-t0):.1f} ms, lml_grad {1e3*(t2-t1):.1f} ms (x{(t2-t1)/(t1-t0):.2f} of a fit), bytes {e.device_bytes()/1e9:.1f} GB, grad {g}")
e.close()
