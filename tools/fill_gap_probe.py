"""How much of the in-situ fill deficit is a clock / power-state ramp?  Fill launches at N separated by idle gaps of
different length, and right after a burst of MFMA work.   python tools/fill_gap_probe.py [N] [kernel]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic
from battgp_amd.engine import ExactGPEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
kname = sys.argv[2] if len(sys.argv) > 2 else "matern32"
kid, hyp = (KERNEL_MATERN32, synthetic.HYP_MATERN32) if kname == "matern32" else (KERNEL_BATTGP, synthetic.HYP_BATTGP)
x, _ = synthetic.make_cell_data(n)
tx = torch.from_numpy(x).cuda()
ld = n + 384
out = torch.empty((n, ld), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
e = ExactGPEngine(kid, hyp)
def fill():
    e.fill_device(tx.data_ptr(), n, tx.data_ptr(), n, 4, out.data_ptr(), ld, lower=1, diag_add=float(hyp[0]))
    t = e.phase_times()
    return t["fill_bytes"] / t["fill_ms"] / 1e6
for _ in range(4): fill()
print(f"{kname} N={n} variant={os.environ.get('BGP_FILL_VARIANT','0')}")
print("steady:", " ".join(f"{fill():.0f}" for _ in range(5)), flush=True)
if os.environ.get("FILL_PROBE_CLOCK"):
    # shader clock granted WHILE the fill runs: warm, and as the first kernel after 100 ms of idle
    import ctypes as C
    nsamp, spin = 600, 3000
    buf = torch.zeros(2 * nsamp, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    def sampled(tag):
        e._lib.bgp_debug_clock_samples_dev(e._h, C.c_void_p(buf.data_ptr()), nsamp, spin)
        r = fill()
        e.sync()
        h = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
        t = (h[:, 0] - h[0, 0]) / 1e5  # ms (100 MHz wall clock)
        pts = []
        for s0 in range(0, nsamp - 30, 30):
            dt = (h[s0 + 30, 0] - h[s0, 0]) / 1e8
            pts.append(f"{t[s0 + 30]:.1f}:{(h[s0 + 30, 1] - h[s0, 1]) / dt / 1e6:.0f}")
        print(f"{tag}: fill {r:.0f} GB/s; t[ms]:sclk[MHz] " + " ".join(pts), flush=True)
    sampled("warm")
    sampled("warm")
    time.sleep(0.1)
    sampled("after 100 ms idle")
    sampled("next")
    def sampled_other(tag, fn):
        torch.cuda.synchronize()
        e._lib.bgp_debug_clock_samples_dev(e._h, C.c_void_p(buf.data_ptr()), nsamp, spin)
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e.sync()
        h = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
        t = (h[:, 0] - h[0, 0]) / 1e5
        pts = []
        for s0 in range(0, nsamp - 30, 30):
            d = (h[s0 + 30, 0] - h[s0, 0]) / 1e8
            pts.append(f"{t[s0 + 30]:.1f}:{(h[s0 + 30, 1] - h[s0, 1]) / d / 1e6:.0f}")
        print(f"{tag} ({dt*1e3:.1f} ms): t[ms]:sclk[MHz] " + " ".join(pts), flush=True)
    nbytes = 4 * n * (n + 1)
    flat = out.view(-1)[: nbytes // 8]
    sampled_other("memset 69 GB", lambda: flat.zero_())
    sampled_other("memset 69 GB x2", lambda: (flat.zero_(), flat.zero_()))
    src = out.view(-1)[nbytes // 8 : nbytes // 8 + nbytes // 16]
    sampled_other("copy 34 GB -> 34 GB", lambda: flat[: nbytes // 16].copy_(src))
    a = torch.randn(16384, 16384, device="cuda", dtype=torch.float64)
    sampled_other("torch fp64 matmul 16384^3", lambda: a @ a)
    sampled_other("elementwise exp() over 4 GB", lambda: torch.exp(flat[: 1 << 29]))
    sys.exit(0)
if os.environ.get("FILL_PROBE_SHORT"):
    time.sleep(0.1)
    print("after 100 ms idle:", f"{fill():.0f} {fill():.0f}")
    sys.exit(0)
nbytes = 4 * n * (n + 1)
def memset():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out.view(-1)[: nbytes // 8].zero_(); b.record(); b.synchronize()
    return nbytes / a.elapsed_time(b) / 1e6
for _ in range(3): memset()
print("memset steady:", " ".join(f"{memset():.0f}" for _ in range(4)))
for gap in (0.005, 0.1):
    time.sleep(gap); r = memset()
    print(f"memset after {gap*1e3:.0f} ms idle: {r:.0f} then {memset():.0f}")
for gap in (0.0002, 0.001, 0.005, 0.02, 0.1, 0.5, 2.0):
    r = []
    for _ in range(3):
        time.sleep(gap)
        r.append(fill())
    print(f"after {gap*1e3:7.1f} ms idle: " + " ".join(f"{v:.0f}" for v in r) + f"   then back-to-back {fill():.0f}")
# after ~0.5 s of MFMA work (a gemm the engine's size)
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
for tag, reps in (("after 0.3 s of fp64 GEMM", 20),):
    torch.cuda.synchronize()
    for _ in range(reps): a @ a
    torch.cuda.synchronize()
    print(tag + ":", f"{fill():.0f} then {fill():.0f}")
e.close()
