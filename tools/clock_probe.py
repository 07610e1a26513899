"""Run the MFMA gemm back-to-back for a few seconds so that rocm-smi can sample the shader clock."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from battgp_amd.engine import ExactGPEngine
from battgp_amd import synthetic
e = ExactGPEngine(0, synthetic.HYP_BATTGP)
m = n = 16384; k = 2048; ld = m + 64
a = torch.randn((k, ld), dtype=torch.float64, device="cuda")
c = torch.randn((n, ld), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 4
it = 0; t0 = time.time()
while time.time() < t_end:
    e.gemm_nt_sub_device(c.data_ptr(), ld, a.data_ptr(), ld, a.data_ptr(), ld, m, n, k, 0); it += 1
dt = time.time() - t0
print(f"{it} gemms, {2.0*m*n*k*it/dt/1e12:.2f} TFLOP/s sustained over {dt:.1f}s")
