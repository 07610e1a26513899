"""Is the in-pipeline fill slower because the chip is hot/throttled right after a 10 s factorisation? (diagnostic)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from battgp_amd import synthetic
from battgp_amd.engine import ExactGPEngine
n = 131072
x, y = synthetic.make_cell_data(n)
e = ExactGPEngine(0, synthetic.HYP_BATTGP)
tx = torch.from_numpy(x).cuda()
ld = n + 384
out = torch.empty((n, ld), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def fills(tag, k=3):
    for i in range(k):
        e.fill_device(tx.data_ptr(), n, tx.data_ptr(), n, 4, out.data_ptr(), ld, lower=1, diag_add=2.33e-6)
        t = e.phase_times()
        print(f"{tag} fill {t['fill_ms']:.2f} ms  {t['fill_bytes']/t['fill_ms']/1e6:.0f} GB/s", flush=True)
fills("idle-start")
e2 = ExactGPEngine(0, synthetic.HYP_BATTGP)
del out; torch.cuda.empty_cache()
e2.fit(x[:100000], y[:100000])   # ~5 s of MFMA
out = torch.empty((n, ld), dtype=torch.float64, device="cuda")
fills("right-after-potrf")
time.sleep(3)
fills("after-3s-sleep")
