"""Workload for the rocprofv3 passes of tools/profile_round.sh: ONE fused fit+predict at N (the bench
default's step) followed by alpha() - whose gemv_t_partial launches read the strictly-lower panels of L
exactly once and calibrate FETCH_SIZE.   python tools/profile_workload.py 40000 [battgp|matern32] [reps]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
kname = sys.argv[2] if len(sys.argv) > 2 else "battgp"
kid, hyp = (KERNEL_MATERN32, synthetic.HYP_MATERN32) if kname == "matern32" else (KERNEL_BATTGP, synthetic.HYP_BATTGP)
x, y = synthetic.make_cell_data(n)
xq = synthetic.make_query(x, 300)
tx, ty, tq = (torch.from_numpy(a).cuda() for a in (x, y, xq))
torch.cuda.synchronize()
eng = ExactGPEngine(kid, hyp, device=0)
if os.environ.get("BGP_LA"):  # look-ahead word (+32: slim chain kernels)
    eng.set_options(lookahead=int(os.environ["BGP_LA"]))
if os.environ.get("BGP_PANEL_SCHEME"):
    eng.set_panel_scheme(int(os.environ["BGP_PANEL_SCHEME"]))
tm = torch.empty(300, dtype=torch.float64, device="cuda")
tv = torch.empty(300, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    eng.fit_predict_device(tx.data_ptr(), ty.data_ptr(), n, 4, tq.data_ptr(), 300, tm.data_ptr(), tv.data_ptr())
mean = tm.cpu().numpy()
ph = eng.phase_times()
a = eng.alpha()
print(json.dumps({"n": n, "kernel": kname, "lml": eng.lml, "phases": ph, "alpha_norm": float(np.linalg.norm(a)), "mean0": float(mean[0])}))
eng.close()
