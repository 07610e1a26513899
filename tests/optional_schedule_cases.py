"""Test bodies for the OPTIONAL Cholesky schedules (lookahead bits 5-7: slim chain kernels, split panels, fused update +
tile Cholesky) - off by default, written while the GPU pool was closed to this build, so they have never met the
hardware.  Not collected by default (no ``test_`` prefix): tests/test_gpu_zz_optional_schedules.py runs every function
here in a CHILD process with a time limit, as the last file of the ``-m gpu`` suite, so that nothing these kernels do on
a real GPU (a fault, a hang, a wrong bit) can cut the core parity record short.  The CPU build of the kernel sources runs
the same functions in the ``-m "not gpu"`` suite (tests/test_emu_kernels.py)."""

import os
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from problem_gen import LA_OPTIONAL, check_problem, problems  # noqa: E402

pytestmark = pytest.mark.gpu

from battgp_amd import synthetic  # noqa: E402
from battgp_amd.engine import ExactGPEngine  # noqa: E402
from oracle import kernels as K  # noqa: E402
from oracle.exact_gp import OracleGP  # noqa: E402


@pytest.mark.parametrize("kid", [K.KERNEL_BATTGP, K.KERNEL_MATERN32])
@pytest.mark.parametrize("n,nb", [(700, 128), (3001, 512), pytest.param(20000, 512, marks=pytest.mark.gpu_sized)])
def test_slim_chain_kernels_and_split_panels_are_bit_identical(kid, n, nb):
    """lookahead bit 5: the diagonal-block chain of every panel after the first runs on the kernels sized to fit next
    to the trailing update's workgroups (potrf_tile_slim, chain_gemm_slim, 32-wide diag_out); bit 6: the panel's solve
    and look-ahead update are split into the next diagonal block's rows (panel stream) and the tall rest (bulk stream);
    bit 7: the rank-64 update of a chain step and the tile Cholesky of the next step share one launch.
    Same arithmetic in the same order for every element, so factor, LML, posterior and tile inverses are identical to
    the last bit - which at N = 20 000 (a trailing update that really saturates the GPU, four streams in flight) is also
    the test of the event graph that orders the streams"""
    hyp = synthetic.HYP_BATTGP if kid == K.KERNEL_BATTGP else synthetic.HYP_MATERN32
    x, y = synthetic.make_cell_data(n, seed=n + 1)
    xq = synthetic.make_query(x, 200)
    out = []
    for la in (1, 1 | 32, 1 | 64, 1 | 32 | 64, 1 | 128, 1 | 64 | 128):
        e = ExactGPEngine(kid, hyp)
        e.set_options(nb_outer=nb, lookahead=la)
        e.set_panel_scheme(1)
        lml, m, v = e.fit_predict(x, y, xq, min_var=-1.0)
        m2, v2 = e.predict(xq[:50], min_var=-1.0)  # later prediction: walks the stored tile / panel inverses
        diag = e.factor_diag()
        res = e.residuals(128)
        e.close()
        assert res[0] < 1e-6 and res[1] < 1e-11, res
        out.append((lml, m, v, m2, v2, diag))
    if n <= 3001:  # every schedule against the oracle, not only against the default schedule
        gp = OracleGP(kid, hyp, x, y).fit()
        m_ref, _ = gp.predict(xq, clamp=False)
        for o in out:
            assert abs(o[0] - gp.lml) <= 1e-6 * abs(gp.lml)
            assert np.linalg.norm(o[1] - m_ref) <= 1e-6 * np.linalg.norm(m_ref)
    for o in out[1:]:
        assert o[0] == out[0][0]
        for a, b in zip(out[0][1:], o[1:]):
            assert np.array_equal(a, b)



@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(problems(n_max=3000, m_max=400, noise_lo=-3.0, la_words=LA_OPTIONAL))
def test_random_problems_match_the_oracle_under_the_optional_schedules(prob):
    """the property of tests/test_gpu_parity.py::test_random_problems_match_the_oracle with the look-ahead words drawn
    from the optional schedules"""
    check_problem(*prob, grad=True)
