"""CPU build of the kernel sources (tests/emu): the UNMODIFIED battgp_amd/csrc/*.hip compiled for the host against a
stand-in <hip/hip_runtime.h> whose workgroups run as cooperative fibers (barriers, readlane / shuffles and the f64
16x16x4 MFMA with the gfx950 lane layout).  These tests run a selection of the ``-m gpu`` parity tests - the very same
test functions, imported from their modules - through that build, at sizes a CPU finishes in seconds: the kernels'
indexing and algebra (fill, tile Cholesky, MFMA GEMM modes, panel schemes, look-ahead, slab layout, triangular solves,
in-place inverse gradient, jitter ladder) are checked against the oracle in the ``-m "not gpu"`` suite as well.

Test infrastructure only: it proves nothing about speed or about what the GPU executes (that is what ``-m gpu`` is for),
the product binding cannot reach this library, and nothing here is ever timed or shipped.
"""

import importlib.util
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_emu_{name}", os.path.join(HERE, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def emu():
    from inject import installed

    with installed() as lib:
        yield lib


@pytest.fixture
def emu_exp(emu):
    """the CPU build of the EXPERIMENTAL library (-DBGP_EXPERIMENTAL: the optional kernel families that await their A/B on the
    hardware) for the duration of ONE test; the module's other tests run on the product's configuration"""
    from inject import installed

    with installed(experimental=True) as lib:
        yield lib


@pytest.fixture(scope="module")
def P(emu):
    return _load("test_gpu_parity")


@pytest.fixture(scope="module")
def S(emu):
    return _load("test_gpu_slab_layout")


def test_cpu_build_exports_the_whole_c_abi(emu):
    from battgp_amd import _lib

    assert emu.bgp_version() >= 100
    for name in _lib.SIGNATURES:
        assert hasattr(emu, name)
    assert os.path.dirname(emu._name).endswith(os.path.join("tests", "emu", "_build"))
    assert _lib.LIB_PATH.endswith(os.path.join("battgp_amd", "libbattgp.so"))  # the product path is untouched


def test_default_library_has_no_optional_schedules_and_says_so(emu):
    """without -DBGP_EXPERIMENTAL the look-ahead bits 5-7 are refused (nothing else of the call is applied)
    and the fill's environment knobs are not read; the experimental configuration accepts the same words"""
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from inject import installed

    e = ExactGPEngine(0, synthetic.HYP_BATTGP)
    for word in (1 | 32, 1 | 64, 1 | 128, 32 | 64 | 128):
        with pytest.raises(RuntimeError, match="experimental build only"):
            e.set_options(nb_outer=128, lookahead=word)
    e.set_options(lookahead=2)  # the measured words still pass
    e.close()
    with installed(experimental=True):
        e = ExactGPEngine(0, synthetic.HYP_BATTGP)
        e.set_options(nb_outer=128, lookahead=1 | 32 | 64 | 128)
        e.close()


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_fill_kernel_vs_oracle(P, kid):
    P.test_kernel_matrix_matches_oracle(kid, 37, 201)
    P.test_kernel_matrix_matches_oracle(kid, 1, 1)


def test_fill_generic_dims_and_unsorted_time(P):
    P.test_kernel_matrix_generic_dims()
    P.test_unsorted_time_column_takes_the_general_wiener_path()


@pytest.mark.parametrize("n", [1, 2, 65, 300])
def test_fit_predict_production_hyperparameters(P, n):
    P.test_fit_predict_battgp_production_hyp(n)


@pytest.mark.parametrize("kid", [1, 2, 3])
def test_fit_predict_other_kernels(P, kid):
    P.test_fit_predict_other_kernels(kid, 200)


def test_reference_known_answers_and_golden_vectors(P, golden_dir):
    P.test_reference_known_answers_on_gpu()
    P.test_stgp_egp_golden_on_gpu(golden_dir)
    P.test_golden_oracle_cases(golden_dir)


def test_jitter_ladder_variance_clamp_refit(P):
    P.test_jitter_ladder_and_not_psd()
    P.test_variance_clamp_matches_gpytorch_min_variance()
    P.test_refit_equals_fresh_fit()
    P.test_fit_predict_with_jitter_retry()


@pytest.mark.parametrize("n,m", [(1, 1), (65, 300), (300, 37)])
def test_fused_equals_separate(P, n, m):
    P.test_fit_predict_fused_equals_separate(n, m)


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_lml_gradient_vs_oracle(emu, kid):
    _load("test_gpu_grad").test_lml_gradient_matches_oracle(kid, 24)


def test_gradient_bookkeeping(emu):
    """phase-time slots around a gradient and the opt-in kept factor (tests/test_gpu_grad.py; the block-bounds test of the
    same module needs torch "device" tensors and runs under `pytest --emu` / on the GPU)"""
    G = _load("test_gpu_grad")
    G.test_restoring_the_factor_does_not_overwrite_the_fits_phase_times()
    G.test_a_failed_restoring_fit_leaves_the_fits_phase_times_alone()
    G.test_keep_factor_brings_the_factor_back_by_a_copy(512)


def test_a_keep_buffer_that_does_not_fit_is_not_asked_for_again(emu, monkeypatch):
    """bgp_set_keep_factor "when memory allows" - when the second factor-sized buffer can not be allocated the
    gradient proceeds without it AND remembers: an optimiser loop (one bgp_lml_grad per iteration) must not repeat a failing
    factor-sized hipMalloc (which also empties the pool of idle handles) every time.  The CPU build's device refuses
    allocations above HIPEMU_MEM_MB; once the bound is lifted the memo still holds, until the switch is set again."""
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K

    x, y = synthetic.make_cell_data(1500, seed=5)
    xq = synthetic.make_query(x, 20)
    e = ExactGPEngine(K.KERNEL_BATTGP, synthetic.HYP_BATTGP)
    try:
        e.fit(x, y)
        m0, _ = e.predict(xq)
        g0 = e.lml_grad()
        e.predict(xq)
        base, factor_bytes = e.device_bytes(), e.layout()[1]
        e.set_keep_factor(True)
        monkeypatch.setenv("HIPEMU_MEM_MB", str(int(factor_bytes / 2**20) - 1))  # the factor-sized buffer no longer fits
        g1 = e.lml_grad()
        assert e.device_bytes() == base and np.array_equal(g0, g1)
        m1, _ = e.predict(xq)  # the factor comes back by the re-run, as without the switch
        assert np.array_equal(m0, m1) and e.phase_times()["restore_ms"] > 0.0
        monkeypatch.delenv("HIPEMU_MEM_MB")
        e.lml_grad()
        assert e.device_bytes() == base  # remembered: not attempted again for a factor of this size
        e.predict(xq)
        e.set_keep_factor(True)  # a new request is a new attempt
        e.lml_grad()
        assert e.device_bytes() - base == factor_bytes
        m2, _ = e.predict(xq)
        assert np.array_equal(m0, m2)
    finally:
        e.close()


@pytest.mark.parametrize("name,n", [("k2", 400), ("k2b", 64), ("k3", 64), ("k1", 10)])
def test_scikit_learn_pins_through_the_kernels(emu, golden_dir, name, n):
    """LML, analytic gradient and latent posterior of the Matern / RBF kernels against the scikit-learn pins"""
    _load("test_gpu_pins_and_sizes").test_matern_and_rbf_match_scikit_learn_pins(golden_dir, name, n)
    _load("test_gpu_grad").test_gradient_matches_scikit_learn_pins(golden_dir, name, n)


@pytest.mark.parametrize("name,n", [("k0prod", 256), ("k0test", 64), ("k1", 10)])
def test_autograd_gradient_pins_through_the_kernels(emu, golden_dir, name, n):
    _load("test_gpu_grad").test_lml_gradient_matches_torch_autograd_pins(golden_dir, name, n)


@pytest.mark.parametrize("experimental", [False, True])
def test_multi_panel_paths_agree_bit_for_bit_and_match_the_oracle(emu, experimental):
    """(experimental: the same on the CPU build of the experimental library, with the optional look-ahead words added.)
    Several outer panels at a CPU-sized N (nb_outer = 128, N = 600): panel schemes 0 and 1, look-ahead off / depth 1 /
    depth 2 / ordered, the column-slab layout, the in-place inverse gradient and the explicit-inverse backward solve - the code
    paths the GPU only reaches from N = 16 384 on with the default widths."""
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    from inject import installed

    with installed(experimental=experimental):
        _multi_panel_body(experimental)


def _multi_panel_body(experimental):
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP, lml_and_grad

    n = 600
    x, y = synthetic.make_cell_data(n, seed=21)
    xq = synthetic.make_query(x, 45)
    for kid, hyp in ((K.KERNEL_BATTGP, synthetic.HYP_BATTGP), (K.KERNEL_MATERN32, synthetic.HYP_MATERN32)):
        ref = OracleGP(kid, hyp, x, y).fit()
        m_ref, v_ref = ref.predict(xq)
        _, g_ref = lml_and_grad(kid, hyp, x, y)
        outs = {}
        for scheme in (0, 1):
            # + 32: slim chain kernels, + 64: split panels, + 128: update + next tile Cholesky in one launch
            las = (0, 1, 2, 9) + (((1 | 128,) + ((1 | 32, 1 | 32 | 64, 1 | 32 | 64 | 128) if scheme == 1 else ())) if experimental else ())
            for la in las:
                e = ExactGPEngine(kid, hyp)
                e.set_options(nb_outer=128, lookahead=la)
                e.set_panel_scheme(scheme)
                lml, mean, var = e.fit_predict(x, y, xq)
                outs[(scheme, la)] = (lml, mean, var)
                if la == 1:
                    g = e.lml_grad()
                    assert np.max(np.abs(g - g_ref) / np.abs(g_ref)) < 1e-7
                    alpha = e.alpha()
                    assert np.linalg.norm(alpha - ref.alpha) < 1e-4 * np.linalg.norm(ref.alpha)
                    res = e.residuals(64)
                    assert res[0] < 1e-7 and res[1] < 1e-12
                e.close()
            base = outs[(scheme, 0)]
            for la in las[1:]:  # look-ahead, slim chain kernels, split panels, fused launches: never a bit
                assert outs[(scheme, la)][0] == base[0]
                assert np.array_equal(outs[(scheme, la)][1], base[1]) and np.array_equal(outs[(scheme, la)][2], base[2])
            assert abs(base[0] - ref.lml) < 1e-9 * abs(ref.lml)
            assert np.linalg.norm(base[1] - m_ref) < 1e-8 * np.linalg.norm(m_ref)
            assert np.max(np.abs(base[2] - v_ref)) < 1e-9 * np.max(np.abs(v_ref))
        # the two schemes order their sums differently: agreement at rounding level
        assert abs(outs[(0, 0)][0] - outs[(1, 0)][0]) < 1e-11 * abs(ref.lml)
        # slab layout: bit-identical to the full square
        e = ExactGPEngine(kid, hyp)
        e.set_options(nb_outer=128)
        e.set_panel_scheme(1)
        e.set_layout(256)
        lml_s, mean_s, var_s = e.fit_predict(x, y, xq)
        g_s = e.lml_grad()
        e.close()
        assert lml_s == outs[(1, 1)][0] and np.array_equal(mean_s, outs[(1, 1)][1]) and np.array_equal(var_s, outs[(1, 1)][2])
        assert np.max(np.abs(g_s - g_ref) / np.abs(g_ref)) < 1e-7


def test_failed_pivot_inside_the_fused_launch_walks_the_jitter_ladder(emu_exp):
    """a pivot that fails inside chain_update_potrf_kernel (lookahead bit 7) must be reported like one that fails in the
    stand-alone tile kernel: same jitter rung, same LML, and NotPSD when the ladder is exhausted"""
    from battgp_amd.engine import ExactGPEngine, NotPSDError, NumericalWarning
    from oracle import kernels as K

    rng = np.random.default_rng(5)
    n = 300
    x = rng.normal(size=(n, 2))
    x[150:] = x[:150]  # every point twice and no noise: singular beyond the first 150 pivots (tiles 2, 3, 4 of a 64-wide chain)
    y = rng.normal(size=n)
    hyp = np.array([0.0, 1.0, 1.0, 1.0])  # ARD-RBF, sigma^2 = 0
    got = []
    for la in (1, 1 | 128):
        for scheme in (0, 1):
            e = ExactGPEngine(K.KERNEL_ARD_RBF, hyp)
            e.set_options(nb_outer=128, lookahead=la)
            e.set_panel_scheme(scheme)
            with pytest.warns(NumericalWarning):
                lml = e.fit(x, y)
            got.append((la, scheme, e.jitter, lml))
            e.set_options(max_tries=0)
            with pytest.raises(NotPSDError):
                e.fit(x, y)
            e.close()
    assert all(g[2] == got[0][2] > 0.0 for g in got), got
    assert got[0][3] == got[2][3] and got[1][3] == got[3][3], got  # fused == unfused, per scheme, to the last bit


def test_slab_layout_and_full_covariance(S):
    S.test_slab_layout_bit_identical_to_full_square(0, 1000, 512, 512, True)
    S.test_slab_width_must_match_the_panel_width()


def test_predict_cov_and_errors(P):
    P.test_errors_are_reported_not_crashed()
    P.test_fit_predict_generic_input_dimension(3)


def test_address_sanitizer_pass():
    """The GPU pool offers no address sanitizer; the CPU build does.  One pass over the engine's code paths (several
    outer panels, both panel schemes, slab layout, later predictions, gradient, backward solve) through the
    AddressSanitizer build, in a child process with the sanitizer runtime preloaded: every global-memory, workspace and
    LDS-array access of the kernel sources is bounds-checked."""
    import subprocess

    import build_emu

    build_emu.build(sanitize="asan", experimental=True)  # the experimental configuration: a superset of the product's code
    rt = build_emu.sanitizer_runtime("asan")
    if not os.path.exists(rt):
        pytest.skip("no shared AddressSanitizer runtime next to the host clang")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", BGP_EMU_EXPERIMENTAL="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "asan_workload.py")], env=env, capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "ASAN-PASS-DONE" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("experimental", [False, True])
def test_stream_graph_under_adversarial_schedules(emu, experimental):
    """The CPU build's streams as real queues (tests/emu/hipemu.cpp): every launch deferred and run in an order that
    honours only in-stream order, event waits and host synchronisation - lazily, eagerly, randomly, and with each of the
    four stream roles (main, panel, copy, bulk) in turn running as far ahead as the graph allows.  All schedules of the
    blocked Cholesky (no look-ahead, depth 1 / 2, ordered, slim chain kernels, split panels) followed by a later
    prediction and the gradient must reproduce the immediate-mode result bit for bit: a missing hipStreamWaitEvent edge
    would not (tests/emu/stream_graph_check.py --mutate drops every wait in turn to show that it is noticed)."""
    import ctypes as C

    import stream_graph_check as G
    from inject import installed

    with installed(experimental=experimental) as lib:
        lib.hipemu_set_sched.argtypes = [C.c_char_p]
        lib.hipemu_set_sched(b"sync")
        try:
            ref = G.workload(1)
            las = (1, 1 | 32 | 64) if experimental else (0, 1, 2, 1 | 8)  # the bulk stream exists for split panels only
            assert all(G.same(G.workload(la), ref) for la in las)
            for policy in ("lazy", "eager", "random:3", "prio:1", "prio:6", "prio:10", "prio:15", "prio:20", "prio:23"):
                assert G.run_policy(lib, policy, las, ref) == [], policy
        finally:
            lib.hipemu_set_sched(b"sync")


@pytest.mark.parametrize("world,sched", [(2, "prio:9"), (3, "lazy")])
def test_sharded_device_backend_ranks_on_the_cpu_build(world, sched):
    """tests/test_gpu_sharded.py's multi-rank worker (DeviceBackend: the engine's per-panel C-ABI building blocks driven
    by sharded.py, gloo between the ranks) with every rank on the CPU build and its streams deferred (HIPEMU_SCHED): the
    packing, offsets, look-ahead exchange and stream hand-offs of the multi-GPU schedule (factorisation, prediction pass and
    the distributed in-place inverse of the gradient) against the oracle, without a GPU"""
    import socket

    import torch.multiprocessing as mp

    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import test_gpu_sharded as G  # by its real name: the spawned ranks unpickle the worker by module

    from battgp_amd import synthetic
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, nb = 450, 64
    saved = {k: os.environ.get(k) for k in ("BGP_TEST_EMU", "HIPEMU_SCHED")}
    os.environ.update(BGP_TEST_EMU="1", HIPEMU_SCHED=sched)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=G._two_rank_worker, args=(r, world, port, n, nb, q)) for r in range(world)]
        [p.start() for p in procs]
        res = sorted(q.get(timeout=400) for _ in range(world))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    x, y = synthetic.make_cell_data(n, seed=9)
    xq = synthetic.make_query(x, 33)
    ref = OracleGP(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y).fit()
    m_ref, v_ref = ref.predict(xq)
    from oracle.exact_gp import lml_and_grad

    _, g_ref = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    for rank, lml, mean, var, grad in res:
        assert abs(lml - ref.lml) < 1e-6 * abs(ref.lml)
        assert np.linalg.norm(np.array(mean) - m_ref) < 1e-6 * np.linalg.norm(m_ref)
        assert np.max(np.abs(np.array(var) - v_ref)) < 1e-9 * synthetic.OUTPUTSCALE_RBF
        # the analytic gradient of the sharded model: Sigma^-1 in place over the ranks' panels, exchanges one step ahead
        assert np.allclose(np.array(grad), g_ref, rtol=1e-5), (rank, grad, g_ref)
    assert all(r[1:] == res[0][1:] for r in res)


def test_optional_interior_paths_of_the_fill():
    """BGP_FILL_MFMA=1 (squared distances of the interior tiles on the matrix pipe: two v_mfma_f64_16x16x4 per 16 x 16
    entries, expanded around the mid-range of the tile's column points) and BGP_FILL_TABLE=256, alone and together,
    against the oracle's kernel code: elementwise relative error of a cross fill with interior tiles below the 2e-13
    parity bound (measured ~1e-14), LML of a training fit with interior tiles below the diagonal at 1e-9, and bits that
    DIFFER from the default path's (the knobs are read once per process: child processes; differing bits prove that
    the optional path was taken and did not fall back)."""
    import json
    import subprocess

    script = os.path.join(HERE, "emu", "fill_variant_check.py")
    envs = {"default": {}, "mfma": {"BGP_FILL_MFMA": "1"}, "mfma+t256": {"BGP_FILL_MFMA": "1", "BGP_FILL_TABLE": "256"}}
    import build_emu

    build_emu.build(experimental=True)  # once, before the children race for it
    procs = {k: subprocess.Popen([sys.executable, script], env=dict(os.environ, BGP_EMU_EXPERIMENTAL="1", **e), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k, e in envs.items()}
    res = {}
    for k, pr in procs.items():
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, se[-2000:]
        res[k] = json.loads(so.strip().splitlines()[-1])
    for k, r in res.items():
        for kid, v in r.items():
            assert v["cross_max_rel"] < 2e-13, (k, kid, v)
            assert v["lml_rel"] < 1e-9, (k, kid, v)
    for k in ("mfma", "mfma+t256"):
        for kid in res[k]:
            if kid != "0":  # K0 takes it only where the sorted-time Wiener form applies: below the diagonal of a training fill
                assert res[k][kid]["digest"] != res["default"][kid]["digest"], (k, kid)            # cross fill took the path
            assert res[k][kid]["factor_digest"] != res["default"][kid]["factor_digest"], (k, kid)  # and so did the training fill


def test_index_arithmetic_beyond_2_pow_32_elements():
    """tests/emu/huge_ld_check.py: small problems stored at a leading dimension of 2^25 (2^23) elements, so that every
    kernel's ``row + col * ld`` passes 2^31 and 2^32 - the sharded device path end to end (fill, factor + pack, updates,
    prediction, in-place inverse, gradient reduction), the MFMA GEMM in all three modes, the block copies and the blocked
    Cholesky driver under every look-ahead word - in seconds, without an N > 46 341 problem"""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "huge_ld_check.py")], env=dict(os.environ, BGP_EMU_EXPERIMENTAL="1"),
                       capture_output=True, text=True, timeout=900)  # experimental configuration: the optional look-ahead words too
    assert r.returncode == 0 and "ok: index arithmetic beyond 2^32 elements" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_lds_poison_reaches_unwritten_shared_arrays():
    """self-test of HIPEMU_POISON=ff (tests/emu/lds_poison_selftest.py): a probe kernel that reads a __shared__ array it
    never wrote sees zeros in the plain CPU build and NaN under the switch - the poisoned runs of the suite are not vacuous"""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "lds_poison_selftest.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok: the LDS poison reaches" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
