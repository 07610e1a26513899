"""Steady fill rate (back-to-back launches on a scratch matrix) for the production kernels at N (default 131072).
    python tools/fill_rate.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic
from battgp_amd.engine import ExactGPEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
x, _ = synthetic.make_cell_data(n)
tx = torch.from_numpy(x).cuda()
ld = n + 384
out = torch.empty((n, ld), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, kid, hyp in (("battgp", KERNEL_BATTGP, synthetic.HYP_BATTGP), ("matern32", KERNEL_MATERN32, synthetic.HYP_MATERN32)):
    e = ExactGPEngine(kid, hyp)
    rates = []
    for i in range(6):
        e.fill_device(tx.data_ptr(), n, tx.data_ptr(), n, 4, out.data_ptr(), ld, lower=1, diag_add=float(hyp[0]))
        t = e.phase_times()
        rates.append(t["fill_bytes"] / t["fill_ms"] / 1e6)
    print(f"{name:9s} N={n}: " + " ".join(f"{r:.0f}" for r in rates) + f"  GB/s   median(2..) {np.median(rates[1:]):.0f}", flush=True)
    e.close()
