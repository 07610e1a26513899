"""TEST INFRASTRUCTURE ONLY: 64-bit index arithmetic of the kernels, checked WITHOUT an N > 46 341 problem.

Every kernel addresses ``row + col * ld``.  At the BASELINE sizes (N = 131 072: 1.7e10 elements) that product passes 2^31
and 2^32; a 32-bit intermediate anywhere would only show there - sizes the CPU build of the kernels cannot run (N^3) and the
GPU has not run for the kernels written since its last contact (in-place inverse, trapezoid reduction, block copies).  The
per-panel C-ABI building blocks take the leading dimension as an argument, so the same index arithmetic is exercised here
with a SMALL problem stored at a HUGE leading dimension (2^25 + 64 elements per column: column 64 starts beyond element
2^31, column 128 beyond 2^32; the buffers are tens of GB of VIRTUAL memory of which only the touched pages exist):

* the sharded driver's whole device path (one rank; fill, factor + pack, rank-nb updates, prediction pass, the distributed
  in-place inverse, gemv_t, trapezoid gradient reduction) with ``PanelLayout.ld`` patched to the huge stride, against the
  oracle: LML, posterior, gradient, restored factor;
* ``bgp_gemm_nt_async_dev`` in all three modes with huge ``lda`` / ``ldb`` / ``ldc`` against numpy (A / B operand reads,
  C read-modify-write, the atomic epilogue);
* ``bgp_block_copy_dev`` plain / transposed / triangular with a huge stride on either side.

    python tests/emu/huge_ld_check.py        (a process of its own: torch's CUDA entry points are faked, inject.py)"""
import ctypes as C
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from inject import fake_cuda_tensors, installed  # noqa: E402

BIG = (1 << 25) + 64  # elements per column: even, 16-byte aligned columns, not a pure power of two


def view(buf, ld, rows, cols):
    """numpy view [rows, cols] (column-major, leading dimension ld) of a torch fp64 buffer"""
    return np.lib.stride_tricks.as_strided(buf.numpy(), shape=(rows, cols), strides=(8, 8 * ld))


def sharded_path():
    from battgp_amd import sharded, synthetic
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    real_ld = sharded.PanelLayout.ld
    sharded.PanelLayout.ld = lambda self, j: BIG + 2 * j  # (also: a different stride per panel)
    try:
        for kid, hyp, n, nb in [(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, 190, 64), (K.KERNEL_MATERN32, synthetic.HYP_MATERN32, 150, 128)]:
            x, y = synthetic.make_cell_data(n, seed=77)
            xq = synthetic.make_query(x, 21)
            ref = OracleGP(kid, hyp, x, y).fit()
            m_ref, v_ref = ref.predict(xq, clamp=False)
            _, g_ref = lml_and_grad(kid, hyp, x, y)
            gp = sharded.make_sharded_gp(kid, hyp, nb=nb)
            try:
                lml = gp.fit(x, y)
                # the store really is addressed beyond 2^31 / 2^32 elements
                last = gp.lay.npanels - 1
                assert gp.poff[last] + gp.lay.ld(last) * (gp.lay.width(last) - 1) > (1 << 31), gp.poff
                assert abs(lml - ref.lml) <= 1e-6 * abs(ref.lml), (lml, ref.lml)
                mean, var = gp.predict(xq, min_var=-1.0)
                assert np.linalg.norm(mean - m_ref) <= 1e-6 * np.linalg.norm(m_ref)
                assert np.max(np.abs(var - v_ref)) <= 1e-7 * hyp[1 if kid else 2]
                g = gp.lml_grad()
                assert np.all(np.abs(g - g_ref) <= 1e-5 * np.maximum(np.abs(g_ref), 1e-3 * np.abs(g_ref).max())), (g, g_ref)
                mean2, var2 = gp.predict(xq, min_var=-1.0)
                assert np.array_equal(mean, mean2) and np.array_equal(var, var2)
                print(f"  sharded path, kernel {kid}, N {n}, nb {nb}, ld {BIG}: ok (last panel at element {gp.poff[last]:.3e})", flush=True)
            finally:
                gp.close()
    finally:
        sharded.PanelLayout.ld = real_ld


def building_blocks():
    from battgp_amd import _lib, synthetic
    from battgp_amd.engine import ExactGPEngine

    lib = _lib.load()
    eng = ExactGPEngine(0, synthetic.HYP_BATTGP)
    h = eng._h
    rng = np.random.default_rng(5)

    def ptr(t, off=0):
        return C.c_void_p(t.data_ptr() + 8 * int(off))

    m, n, k = 200, 138, 128  # (even m, n; k a multiple of 16; column 127 of A / B sits beyond element 2^32)
    a_buf, b_buf, c_buf = (torch.empty(BIG * cols + 1024, dtype=torch.float64) for cols in (k, k, n))
    A, B, Cm = view(a_buf, BIG, m, k), view(b_buf, BIG, n, k), view(c_buf, BIG, m, n)
    for mode in (0, 1, 2):
        A[:] = rng.normal(size=(m, k))
        B[:] = rng.normal(size=(n, k))
        Cm[:] = c0 = rng.normal(size=(m, n))
        rc = lib.bgp_gemm_nt_async_dev(h, mode, ptr(c_buf), BIG, ptr(a_buf), BIG, ptr(b_buf), BIG, m, n, k, 0, 0)
        assert rc == 0, lib.bgp_last_error(h)
        assert lib.bgp_sync(h) == 0
        prod = A @ B.T
        want = {0: c0 - prod, 1: prod, 2: c0 - prod}[mode]
        err = np.max(np.abs(Cm - want)) / np.max(np.abs(want))
        assert err < 1e-13, (mode, err)
        print(f"  gemm_nt mode {mode}, lda = ldb = ldc = {BIG}: ok ({err:.1e})", flush=True)
    # block copies: huge stride on the source, on the destination, transposed, triangular
    rows, cols = 150, 131
    s_buf, d_buf = torch.empty(BIG * cols + 1024, dtype=torch.float64), torch.empty(BIG * rows + 1024, dtype=torch.float64)
    S = view(s_buf, BIG, rows, cols)
    S[:] = src = rng.normal(size=(rows, cols))
    for trans, tri, scale in [(0, 0, 1.0), (1, 0, -1.0), (1, 1, 2.0), (0, 1, 1.0)]:
        D = view(d_buf, BIG, cols if trans else rows, rows if trans else cols)
        D[:] = 7.0
        rc = lib.bgp_block_copy_dev(h, ptr(s_buf), BIG, rows, cols, ptr(d_buf), BIG, trans, scale, tri)
        assert rc == 0 and lib.bgp_sync(h) == 0, lib.bgp_last_error(h)
        want = scale * (np.tril(src) if tri else src)
        want = want.T if trans else want
        assert np.array_equal(D, want), (trans, tri)
        print(f"  block_copy trans {trans} tri {tri}, lds = ldd = {BIG}: ok", flush=True)
    # the engine's own blocked Cholesky driver (panel chain, look-ahead, rank-NB trailing updates by atomics) on a matrix
    # stored at a huge leading dimension: columns 256.. lie beyond element 2^31, 512.. beyond 2^32
    lda, nn = (1 << 23) + 64, 576
    m_buf = torch.empty(lda * nn + 1024, dtype=torch.float64)
    M = view(m_buf, lda, nn, nn)
    g = rng.normal(size=(nn, nn))
    spd = g @ g.T + nn * np.eye(nn)
    for la, scheme, nb in [(1, 0, 128), (1, 1, 128), (2, 1, 64), (0, 0, 256), (1 | 32, 1, 128), (1 | 32 | 64, 1, 128), (1 | 128, 1, 128)]:
        M[:] = spd
        eng.set_options(nb_outer=nb, lookahead=la)
        eng.set_panel_scheme(scheme)
        info = C.c_int(-1)
        rc = lib.bgp_potrf_dev(h, ptr(m_buf), nn, lda, C.byref(info))
        assert rc == 0 and info.value == 0, (rc, info.value, lib.bgp_last_error(h))
        err = np.max(np.abs(np.tril(M) - np.linalg.cholesky(spd))) / np.sqrt(nn)
        assert err < 1e-13, (la, scheme, nb, err)
        print(f"  potrf driver lookahead {la} scheme {scheme} nb {nb}, n = {nn}, lda = {lda}: ok ({err:.1e})", flush=True)
    eng.close()


def engine_path():
    """the single-GPU engine end to end with ``bgp_debug_set_ld_pad``: fit / fused fit+predict (query rows riding below the
    matrix) / later predict / full covariance / residual checks / in-place-inverse gradient / refit, full square and
    column slabs, at a leading dimension of ~2^23: the columns from 256 on lie beyond element 2^31, from 512 on beyond 2^32"""
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    pad = 1 << 23
    for kid, hyp, n, nb, slab, scheme in [(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, 600, 128, -1, 1), (K.KERNEL_MATERN32, synthetic.HYP_MATERN32, 450, 64, 128, 0),
                                          (K.KERNEL_BATTGP, synthetic.HYP_BATTGP, 330, 128, 128, 1)]:
        x, y = synthetic.make_cell_data(n, seed=88)
        xq = synthetic.make_query(x, 37)
        ref = OracleGP(kid, hyp, x, y).fit()
        m_ref, v_ref = ref.predict(xq, clamp=False)
        _, c_ref = ref.predict(xq[:9], full_cov=True)
        _, g_ref = lml_and_grad(kid, hyp, x, y)
        prior = hyp[1 if kid else 2]
        e = ExactGPEngine(kid, hyp)
        try:
            e.set_options(nb_outer=nb, lookahead=1)
            e.set_panel_scheme(scheme)
            e.set_layout(slab)
            e.debug_set_ld_pad(pad)
            lml, mean, var = e.fit_predict(x, y, xq, min_var=-1.0)
            width, nbytes = e.layout()
            assert nbytes > 8 * (1 << 31), nbytes  # the buffer really spans more than 2^31 (N = 600: 2^32) elements
            assert (width > 0) == (slab > 0)
            lml_b = e.fit(x, y)
            mean_b, var_b = e.predict(xq, min_var=-1.0)
            for ll, mm, vv in ((lml, mean, var), (lml_b, mean_b, var_b)):
                assert abs(ll - ref.lml) <= 1e-6 * abs(ref.lml), (ll, ref.lml)
                assert np.linalg.norm(mm - m_ref) <= 1e-6 * np.linalg.norm(m_ref)
                assert np.max(np.abs(vv - v_ref)) <= 1e-7 * prior
            _, cov = e.predict_cov(xq[:9])
            assert np.max(np.abs(cov - c_ref)) <= 1e-7 * prior
            r_solve, r_llt = e.residuals(64)
            assert r_solve < 1e-6 and r_llt < 1e-12, (r_solve, r_llt)
            g = e.lml_grad()
            assert np.all(np.abs(g - g_ref) <= 1e-5 * np.maximum(np.abs(g_ref), 1e-3 * np.abs(g_ref).max())), (g, g_ref)
            mean_c, var_c = e.predict(xq, min_var=-1.0)  # the factor the gradient consumed is rebuilt
            assert np.array_equal(mean_c, mean_b) and np.array_equal(var_c, var_b)
            hyp2 = hyp.copy()
            hyp2[1] *= 1.5
            assert abs(e.refit(hyp2) - OracleGP(kid, hyp2, x, y).fit().lml) <= 1e-6 * abs(ref.lml)
            print(f"  engine path, kernel {kid}, N {n}, nb {nb}, slab {slab}, scheme {scheme}, ld pad 2^23 ({nbytes / 2**30:.0f} GiB virtual): ok", flush=True)
        finally:
            e.close()


fake_cuda_tensors()
os.environ.setdefault("HIPEMU_MEM_GB", "100")  # the emulated device refuses allocations above this (virtual memory here)
# --ubsan: the UndefinedBehaviorSanitizer build (signed overflow of a 32-bit intermediate aborts even where the wrapped value
# would happen to be harmless)
with installed(sanitize="--ubsan" in sys.argv), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    building_blocks()
    sharded_path()
    engine_path()
    print("ok: index arithmetic beyond 2^32 elements", flush=True)
