"""The C-ABI from C, on the GPU (ordered after the Python-driven parity modules: tests/conftest.py)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from battgp_amd import synthetic  # noqa: E402
from oracle import kernels as K  # noqa: E402


def test_plain_c_caller_on_the_gpu(tmp_path, request):
    """tests/cabi/c_caller.c - a C99 program that includes include/battgp.h and links libbattgp.so - on the GPU: the
    reference's known answers, argument errors, and LML + analytic gradient of a production-kernel problem against
    the oracle (the same program the CPU suite runs against the CPU build of the kernel sources)."""
    import shutil
    import subprocess

    from battgp_amd import _lib
    from oracle.exact_gp import lml_and_grad

    if request.config.getoption("--emu"):
        pytest.skip("links the PRODUCT library: needs the GPU (the CPU suite runs the same program against the CPU build)")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_caller")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cabi", "c_caller.c"), "-L" + libdir, "-lbattgp", "-lm", "-o", exe], check=True, capture_output=True)
    n, d = 700, 4
    x, y = synthetic.make_cell_data(n, seed=31)
    lml, grad = lml_and_grad(K.KERNEL_BATTGP, synthetic.HYP_BATTGP, x, y)
    args = [str(n), str(d), repr(float(lml))] + [repr(float(v)) for v in x.ravel()] + [repr(float(v)) for v in y] + [repr(float(v)) for v in synthetic.HYP_BATTGP]
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c_caller ok" in r.stdout, r.stdout + r.stderr
    got = np.array([float(v) for v in r.stdout.splitlines()[0].split("grad")[1].split()])
    assert np.all(np.abs(got - grad) <= 1e-5 * np.abs(grad) + 1e-7 * np.abs(grad).max()), (got, grad)
