"""TEST INFRASTRUCTURE ONLY: the optional interior paths of the covariance fill (BGP_FILL_MFMA=1: squared distances on
the matrix pipe; BGP_FILL_TABLE=256) on the CPU build of the kernel sources, against the oracle's kernel code.  The
knobs are read once per process, so tests/test_emu_kernels.py runs this script in child processes with and without them
and compares: elementwise error bound in both, and DIFFERENT bits (the optional path was really taken).
With --gpu the shipped library on the real device is checked instead of the CPU build.
Prints one JSON line: per kernel the largest elementwise relative error of a cross fill with interior tiles, of a
training fit (LML) with interior tiles below the diagonal, and a digest of the matrices."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import contextlib  # noqa: E402

import numpy as np  # noqa: E402

if "--gpu" in sys.argv:  # the same check through the shipped library on a real GPU (tests/test_gpu_zz_optional_schedules.py)
    sys.argv.remove("--gpu")
    installed = contextlib.nullcontext
else:
    from inject import installed  # noqa: E402

with installed():
    from battgp_amd import synthetic
    from battgp_amd.engine import ExactGPEngine
    from oracle import kernels as K
    from oracle.exact_gp import OracleGP

    out = {}
    hyps = {0: synthetic.HYP_BATTGP, 1: np.array([2.33e-6, 0.0099, 300.0]), 2: synthetic.HYP_MATERN32,
            3: np.array([2.33e-6, 0.0099, 400.0, 12.11, 33.75, 45.14])}
    for kid, hyp in hyps.items():
        x, y = synthetic.make_cell_data(1100, seed=11)     # rows: two full 512-row tiles + a ragged one
        x2 = synthetic.make_cell_data(96, seed=12)[0]      # columns: three full 32-column tiles
        x2[5] = x[700]                                     # a coincident pair inside an interior tile
        e = ExactGPEngine(kid, hyp)
        got = e.kernel_matrix(x2, x)                       # the second argument's points are the rows of the fill
        ref = K.kernel_matrix(kid, hyp, x2, x)
        rel = float(np.max(np.abs(got - ref) / np.abs(ref)))
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 1100  # training fill: interior tiles strictly below the diagonal
        xt, yt = synthetic.make_cell_data(n, seed=13)
        lml = e.fit(xt, yt)
        lml_ref = OracleGP(kid, hyp, xt, yt).fit().lml
        lrow = e.factor_rows([n - 1])  # last row of L: depends on every entry of the filled triangle
        e.close()
        out[str(kid)] = {"cross_max_rel": rel, "lml_rel": abs(lml - lml_ref) / abs(lml_ref),
                         "digest": hashlib.sha256(got.tobytes()).hexdigest()[:16], "lml": lml,
                         "factor_digest": hashlib.sha256(lrow.tobytes()).hexdigest()[:16]}
    print(json.dumps(out))
