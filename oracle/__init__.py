"""CPU oracle for the BattGP ``full_gp`` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``battgp_amd`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

The reference's arithmetic for this path lives in third-party, un-vendored code
(``gpytorch>=1.11`` / ``linear_operator`` / ``torch``, see
``/root/reference/requirements.txt:15-16``) that is not installed in the build
container, so this oracle is a numpy/scipy-LAPACK *restatement* of the
exact-Cholesky branch of that algebra, anchored on the reference's own call
sites (file:line cited per function) and pinned by

* the reference's closed-form known-answer tests
  (``tests/gp/test_standard_models.py:32-33,46-47``,
  ``tests/gp/test_recursive_gp.py:85-102``), and
* the reference's ``test_compare_stgp_egp``
  (``tests/gp/test_spatiotemporal_gp.py:218-282``): exact GP with the
  production kernel family == Kalman spatio-temporal GP at 1e-6 rel, where the
  Kalman side is driven by the reference's own importable, pure-numpy
  ``WienerTemporalKernel`` (``src/gp/wiener_kernel_temporal.py:28-35``).
  Golden vectors produced that way are committed under ``tests/golden/``
  together with the generating script.

Parity status: K0 (Wiener+ARD-RBF) and K1 (scaled RBF) posterior mean/variance
are pinned as above.  No reference test pins an LML value, jitter behaviour or
the Matern-3/2 kernel (the reference has no Matern call site at all): for those
the oracle is "parity unpinned" by the reference and rests on textbook algebra
plus self-consistency checks.
"""

from .kernels import (  # noqa: F401
    KERNEL_BATTGP,
    KERNEL_MATERN32,
    KERNEL_SCALED_RBF,
    kernel_diag,
    kernel_matrix,
    n_hyp,
)
from .exact_gp import (  # noqa: F401
    NotPSDError,
    OracleGP,
    lml_and_grad,
    psd_safe_cholesky,
)
