"""Standalone timing of the MFMA gemm_nt kernel through the C-ABI (diagnostic, not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from battgp_amd.engine import ExactGPEngine
from battgp_amd import synthetic

e = ExactGPEngine(0, synthetic.HYP_BATTGP)
for (m, n, k, lower) in [(16384, 16384, 512, 0), (16384, 16384, 2048, 0), (16384, 16384, 64, 0), (32768, 32768, 512, 1), (16384, 512, 512, 0)]:
    ld = m + 64
    a = torch.randn((k, ld), dtype=torch.float64, device="cuda")
    c = torch.randn((n, ld), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        e.gemm_nt_sub_device(c.data_ptr(), ld, a.data_ptr(), ld, a.data_ptr(), ld, m, n, k, lower)
        dt = time.perf_counter() - t0
        best = min(best, dt)
    flop = 2.0 * m * n * k * (0.5 if lower else 1.0)
    print(f"gemm_nt m={m} n={n} k={k} lower={lower}: {best*1e3:8.3f} ms  {flop/best/1e12:6.2f} TFLOP/s", flush=True)
    del a, c
e.close()
