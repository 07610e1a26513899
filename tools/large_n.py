"""One fit+predict at a size beyond the full-square ceiling (N > ~196 000 on 288 GB): the factor lives
in column slabs (bgp_set_layout).  Prints one JSON line with timings, layout and on-device residuals.

    python tools/large_n.py 262144 [battgp|matern32] [slab_width: 0 auto, -1 full square, >0 width] [m] [nb_outer] [panel scheme]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

n = int(sys.argv[1])
kernel = sys.argv[2] if len(sys.argv) > 2 else "battgp"
slab = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m = int(sys.argv[4]) if len(sys.argv) > 4 else 300
kid, hyp = (KERNEL_BATTGP, synthetic.HYP_BATTGP) if kernel == "battgp" else (KERNEL_MATERN32, synthetic.HYP_MATERN32)

x, y = synthetic.make_cell_data(n)
xq = synthetic.make_query(x, m)
free0, total = torch.cuda.mem_get_info()
eng = ExactGPEngine(kid, hyp, device=0)
eng.set_layout(slab)
nb = int(sys.argv[5]) if len(sys.argv) > 5 else -1
if nb > 0:
    eng.set_options(nb_outer=nb)
if len(sys.argv) > 6:
    eng.set_panel_scheme(int(sys.argv[6]))
t0 = time.perf_counter()
lml, mean, var = eng.fit_predict(x, y, xq)
wall = time.perf_counter() - t0
ph = eng.phase_times()
width, fbytes = eng.layout()
res = eng.residuals(256)
flop = n**3 / 3.0 + float(n) * n * m + 2.0 * n * n
out = {
    "n": n, "m": m, "kernel": kernel, "nb_outer": nb if nb > 0 else "auto", "trail_launches": ph["trail_launches"], "slab_width": width, "factor_bytes": fbytes, "device_bytes": eng.device_bytes(),
    "hbm_total": total, "hbm_free_before": free0, "full_square_bytes": 8 * (n + 384) * n,
    "fit_predict_s": wall, "gflops": flop / wall / 1e9,
    "potrf_tflops": (n**3 / 3.0) / (ph["potrf_ms"] * 1e-3) / 1e12,
    "trail_tflops": ph["trail_flop"] / (ph["trail_ms"] * 1e-3) / 1e12,
    "fill_gbs": ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9,
    "phases_ms": {k: ph[k] for k in ("fill_ms", "potrf_ms", "solve_ms", "cross_ms", "var_ms", "trail_ms")},
    "lml": lml, "jitter": eng.jitter, "residuals": {"rel_solve": res[0], "max_llt": res[1]},
    "mean_first": [float(v) for v in mean[:3]], "var_first": [float(v) for v in var[:3]],
}
eng.close()
print(json.dumps(out), flush=True)
