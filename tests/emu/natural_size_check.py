"""TEST INFRASTRUCTURE ONLY (long-running, not part of the CPU suite): the engine's AUTOMATIC defaults at the sizes where they
switch code paths - panel scheme 1 from N = 16 384, outer panel width 1024 from N = 32 768 - through the CPU build of the
kernel sources, on a ragged N just above the threshold: fused fit + predict, a later predict, the in-place-inverse gradient
and the rebuilt factor, against the oracle (LAPACK).  The CPU build runs at 2-3 GFLOP/s: N = 16 400 takes ~ 1/2 hour with
the gradient, N = 32 800 a few hours.

    HIPEMU_MEM_GB=24 python tests/emu/natural_size_check.py 16400 [--no-grad]"""
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
from inject import installed  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16400
want_grad = "--no-grad" not in sys.argv
# --engine-grad g0,g1,...: skip the engine (hours at N = 32 800) and compare the oracle's gradient with these printed values
given = next((np.array([float(v) for v in a.split("=", 1)[1].split(",")]) for a in sys.argv if a.startswith("--engine-grad=")), None)

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402

with installed(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, 300)
    hyp = synthetic.HYP_BATTGP
    t0 = time.time()
    ref = OracleGP(K.KERNEL_BATTGP, hyp, x, y).fit()
    m_ref, v_ref = ref.predict(xq, clamp=False)
    print(f"oracle N={n}: {time.time() - t0:.0f} s, lml {ref.lml:.6f}", flush=True)
    e = ExactGPEngine(K.KERNEL_BATTGP, hyp)  # no option set: scheme, panel width and layout are the automatic ones
    try:
        if given is not None:
            raise StopIteration
        t0 = time.time()
        lml, mean, var = e.fit_predict(x, y, xq, min_var=-1.0)
        print(f"engine fit+predict: {time.time() - t0:.0f} s, lml rel {abs(lml - ref.lml) / abs(ref.lml):.1e}, "
              f"mean rel {np.linalg.norm(mean - m_ref) / np.linalg.norm(m_ref):.1e}, var abs/prior {np.max(np.abs(var - v_ref)) / hyp[2]:.1e}", flush=True)
        assert abs(lml - ref.lml) <= 1e-6 * abs(ref.lml)
        assert np.linalg.norm(mean - m_ref) <= 1e-6 * np.linalg.norm(m_ref)
        assert np.max(np.abs(var - v_ref)) <= 1e-7 * hyp[2]
        mean2, var2 = e.predict(xq, min_var=-1.0)
        assert np.linalg.norm(mean2 - m_ref) <= 1e-6 * np.linalg.norm(m_ref) and np.max(np.abs(var2 - v_ref)) <= 1e-7 * hyp[2]
        print("later predict (separate solve pass): ok", flush=True)
        if want_grad:
            t0 = time.time()
            g = e.lml_grad()
            print(f"engine gradient: {time.time() - t0:.0f} s {g}", flush=True)
            # oracle gradient without forming the N x N derivative matrices at once: 1/2 tr((alpha alpha^T - Sigma^-1) dK)
            import torch  # (this image's OpenBLAS - scipy's and numpy's - segfaults in level-3 routines from N = 32 768 on: MKL)

            from oracle.exact_gp import kernel_derivatives

            t0 = time.time()
            linv = torch.linalg.solve_triangular(torch.from_numpy(ref.L), torch.eye(n, dtype=torch.float64), upper=False)
            w = -(linv.T @ linv)
            del linv
            w = w.numpy()
            for i0 in range(0, n, 4096):  # + alpha alpha^T, block-wise (elementwise only)
                w[i0:i0 + 4096] += ref.alpha[i0:i0 + 4096, None] * ref.alpha[None, :]
            g_ref = np.zeros(hyp.size)
            g_ref[0] = 0.5 * np.trace(w)
            blk = 2048
            for i0 in range(0, n, blk):
                dks = kernel_derivatives(K.KERNEL_BATTGP, hyp, x[i0:i0 + blk], x)
                for i, dk in enumerate(dks):
                    g_ref[1 + i] += 0.5 * np.sum(w[i0:i0 + blk] * dk)
            print(f"oracle gradient: {time.time() - t0:.0f} s {g_ref}", flush=True)
            rel = np.abs(g - g_ref) / np.abs(g_ref)  # every component relative to ITSELF (they span 12 decades)
            print("gradient rel", rel, flush=True)
            assert np.all(rel <= 1e-6)
            mean3, var3 = e.predict(xq, min_var=-1.0)  # the factor comes back bit for bit
            assert np.array_equal(mean3, mean2) and np.array_equal(var3, var2)
            print("factor restored after the gradient: ok", flush=True)
    except StopIteration:
        import torch

        from oracle.exact_gp import kernel_derivatives

        t0 = time.time()
        linv = torch.linalg.solve_triangular(torch.from_numpy(ref.L), torch.eye(n, dtype=torch.float64), upper=False)
        w = -(linv.T @ linv)
        del linv
        w = w.numpy()
        for i0 in range(0, n, 4096):
            w[i0:i0 + 4096] += ref.alpha[i0:i0 + 4096, None] * ref.alpha[None, :]
        g_ref = np.zeros(hyp.size)
        g_ref[0] = 0.5 * np.trace(w)
        for i0 in range(0, n, 2048):
            for i, dk in enumerate(kernel_derivatives(K.KERNEL_BATTGP, hyp, x[i0:i0 + 2048], x)):
                g_ref[1 + i] += 0.5 * np.sum(w[i0:i0 + 2048] * dk)
        rel = np.abs(given - g_ref) / np.abs(g_ref)
        print(f"oracle gradient: {time.time() - t0:.0f} s {g_ref}\nengine (as printed by the earlier run) vs oracle, per component: {rel}", flush=True)
        assert np.all(rel <= 1e-6)
    finally:
        e.close()
    print(f"ok: automatic defaults at N = {n}", flush=True)
