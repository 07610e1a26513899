"""TEST INFRASTRUCTURE ONLY: one pass over the engine's code paths through the AddressSanitizer build of the CPU
stand-in (tests/emu).  Run by tests/test_emu_kernels.py::test_address_sanitizer_pass in a child process that has the
sanitizer runtime preloaded; every global / LDS / workspace access of the UNMODIFIED kernel sources is bounds-checked
(device buffers are plain heap allocations there, LDS arrays are stack/static arrays of the block).  Covers the single-GPU engine
(fit, fused and later predictions, gradient, slab layout, optional schedules) and the sharded driver's building blocks."""

import os
import sys

os.environ["BGP_FILL_TABLE"] = "256"  # this process also takes the optional 256-entry interior table of the fill
os.environ["BGP_FILL_MFMA"] = "1"     # ... and the matrix-pipe form of the interior tiles' squared distances

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
from inject import installed  # noqa: E402


def main() -> int:
    with installed(sanitize="asan"):
        from battgp_amd import synthetic
        from battgp_amd.engine import ExactGPEngine

        for kid, hyp in ((0, synthetic.HYP_BATTGP), (2, synthetic.HYP_MATERN32)):
            cases = ((1, 1, 0, 0, 1, 0), (65, 3, 0, 1, 1, 0), (333, 70, 128, 1, 1, 0), (333, 70, 128, 0, 2, 0), (400, 9, 128, 1, 1, 256), (333, 5, 128, 1, 1 | 32 | 64, 0), (333, 5, 128, 0, 1 | 128, 0))
            for n, m, nb, scheme, la, slab in cases if kid == 0 else (cases[2], cases[5], cases[6]):  # cases[5]: slim chain kernels + split panels, cases[6]: fused update + potrf
                x, y = synthetic.make_cell_data(n, seed=5)
                xq = synthetic.make_query(x, m)
                e = ExactGPEngine(kid, hyp)
                if nb:
                    e.set_options(nb_outer=nb, lookahead=la)
                e.set_panel_scheme(scheme)
                if slab:
                    e.set_layout(slab)
                lml, mean, var = e.fit_predict(x, y, xq)
                assert np.isfinite(lml) and np.all(np.isfinite(mean)) and np.all(var > 0)
                m2, v2 = e.predict(synthetic.make_query(x, m + 5))  # later prediction: separate query pass
                assert np.all(np.isfinite(m2)) and np.all(v2 > 0)
                g = e.lml_grad()
                assert np.all(np.isfinite(g))
                a = e.alpha()
                assert np.all(np.isfinite(a))
                r = e.residuals(16)
                assert r[0] < 1e-6
                e.refit(hyp)
                e.close()
        # interior tiles of the fill (512 x 32, off the diagonal): cross fills with two full row tiles and a ragged third -
        # the optional interior paths set above (the sorted-time K0 form below the diagonal of a training fill is covered,
        # without the sanitizer, by tests/test_emu_kernels.py::test_optional_interior_paths_of_the_fill)
        xs, ys = synthetic.make_cell_data(1100, seed=11)
        for kid, hyp in ((2, synthetic.HYP_MATERN32), (3, np.array([2.33e-6, 0.0099, 400.0, 12.11, 33.75, 45.14]))):
            e = ExactGPEngine(kid, hyp)
            km = e.kernel_matrix(synthetic.make_cell_data(64, seed=12)[0], xs)
            assert km.shape == (64, 1100) and np.all(np.isfinite(km))
            e.close()
        # the sharded driver's per-panel building blocks (fill block, factor + pack, panel updates, panel solve, row
        # dots) on one rank: ragged sizes, several panels, a single short panel
        from inject import fake_cuda_tensors

        fake_cuda_tensors()  # this child process only: the driver's torch "device" buffers are host memory here
        from battgp_amd.sharded import make_sharded_gp

        for kid, hyp, n, nb in ((0, synthetic.HYP_BATTGP, 333, 128), (2, synthetic.HYP_MATERN32, 61, 64)):
            x, y = synthetic.make_cell_data(n, seed=3)
            gp = make_sharded_gp(kid, hyp, nb=nb, backend_name="gloo", local_rank=0)
            lml = gp.fit(x, y)
            mean, var = gp.predict(synthetic.make_query(x, 37))
            assert np.isfinite(lml) and np.all(np.isfinite(mean)) and np.all(var > 0)
            g = gp.lml_grad()  # Sigma^-1 in place over the panels: block copies, the general GEMM entry, panel inverse, gemv_t, reduction
            assert np.all(np.isfinite(g))
            mean2, _ = gp.predict(synthetic.make_query(x, 37))  # the factor comes back
            assert np.array_equal(mean, mean2)
            gp.close()
    print("ASAN-PASS-DONE")
    return 0


if __name__ == "__main__":
    sys.exit(main())
