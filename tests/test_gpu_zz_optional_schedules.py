"""The OPTIONAL Cholesky schedules (lookahead bits 5-7) on the GPU - last file of the ``-m gpu`` suite, every case in a
CHILD process with a time limit.

These kernels were written while the GPU pool was closed to this build: they are off by default, bit-identical to the
default schedule in the CPU build of the kernel sources, and have never run on the hardware.  Whatever they do there -
a memory fault, a hang, a differing bit - must fail THIS test only, after every core parity test has already been
recorded; a fault inside the pytest process itself would take the whole record down."""

import os
import subprocess
import sys

import pytest

# Off-by-default code that has not met the hardware yet does not run unless asked for (BGP_TEST_OPTIONAL=1; `pytest --emu`
# asks by itself): a kernel that hangs the GPU cannot be recovered by killing the child, and the driver's round-end
# sequence runs smoke() and bench.py on the same box right after this suite - an experiment must not be able to cost the
# headline record.  tools/gpu_session.sh runs them in its own last stage ("optional"), after everything else of the
# session has been written out.  When they DO run, a failure is a failure (no xfail): the stage goes red, and
# tools/decide_ab.py reads its log and refuses to promote a variant whose parity case did not pass.
# They live in the EXPERIMENTAL library only (battgp_amd/build.py --experimental, -DBGP_EXPERIMENTAL): on the GPU the stage
# sets BGP_EXPERIMENTAL_LIB=1 (battgp_amd/_lib.py then loads libbattgp_exp.so), under `pytest --emu` BGP_EMU_EXPERIMENTAL=1
# selects the CPU build of the same configuration.
_EMU = "--emu" in sys.argv
_ASKED = (os.environ.get("BGP_EMU_EXPERIMENTAL") == "1") if _EMU else (os.environ.get("BGP_TEST_OPTIONAL") == "1" and os.environ.get("BGP_EXPERIMENTAL_LIB") == "1")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _ASKED, reason="optional schedules of the experimental library: BGP_TEST_OPTIONAL=1 BGP_EXPERIMENTAL_LIB=1 on the GPU "
                                                    "(tools/gpu_session.sh stage 'optional'), BGP_EMU_EXPERIMENTAL=1 with --emu")]

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.path.join(HERE, "optional_schedule_cases.py")


@pytest.mark.parametrize("case,limit_s", [
    ("test_slim_chain_kernels_and_split_panels_are_bit_identical", 420),
    ("test_random_problems_match_the_oracle_under_the_optional_schedules", 300),
])
def test_optional_schedules_in_a_child_process(request, case, limit_s):
    cmd = [sys.executable, "-m", "pytest", f"{CASES}::{case}", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    if request.config.getoption("--emu"):
        cmd += ["--emu"]  # (the N = 20 000 case is marked gpu_sized: hours on the CPU build, skipped there by tests/conftest.py)
    try:
        r = subprocess.run(cmd, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=limit_s)
    except subprocess.TimeoutExpired as exc:
        pytest.fail(f"{case}: no result within {limit_s} s (child killed); stdout tail: {(exc.stdout or b'')[-800:]!r}")
    assert r.returncode == 0, f"{case}: child rc = {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}"


@pytest.mark.skipif(_EMU, reason="drives the shipped library on a real GPU; the CPU build's check is tests/test_emu_kernels.py::test_optional_interior_paths_of_the_fill")
def test_optional_interior_paths_of_the_fill_on_the_gpu():
    """BGP_FILL_MFMA=1 / BGP_FILL_TABLE=256 (read once per process: child processes) - the check of
    tests/test_emu_kernels.py::test_optional_interior_paths_of_the_fill through the shipped library: elementwise error
    of the filled entries against the oracle's kernel code, LML of a training fit, and bits that differ from the default
    path's (the optional path was taken)."""
    import json

    script = os.path.join(HERE, "emu", "fill_variant_check.py")
    envs = {"default": {}, "mfma": {"BGP_FILL_MFMA": "1"}, "mfma+t256": {"BGP_FILL_MFMA": "1", "BGP_FILL_TABLE": "256"}, "t256": {"BGP_FILL_TABLE": "256"}}
    res = {}
    for k, e in envs.items():
        try:
            r = subprocess.run([sys.executable, script, "--gpu", "3000"], env=dict(os.environ, **e), capture_output=True, text=True, timeout=180)
        except subprocess.TimeoutExpired:
            pytest.fail(f"fill variant {k}: no result within 180 s")
        assert r.returncode == 0, f"fill variant {k}: rc = {r.returncode}\n{r.stderr[-2000:]}"
        res[k] = json.loads(r.stdout.strip().splitlines()[-1])
    for k, rr in res.items():
        for kid, v in rr.items():
            assert v["cross_max_rel"] < 2e-13 and v["lml_rel"] < 1e-9, (k, kid, v)
    for k in ("mfma", "mfma+t256", "t256"):
        for kid in res[k]:
            assert res[k][kid]["factor_digest"] != res["default"][kid]["factor_digest"], (k, kid)
