"""Property test of the whole C-ABI path on the CPU build of the kernel sources (tests/emu): random sizes (ragged: N, M
not multiples of anything), input dimensions 1..6, all four kernels, random hyper-parameters over decades, unsorted /
duplicated time stamps, coincident points, both panel schemes, every look-ahead word, the column-slab layout - LML,
posterior mean and variance against the oracle, and (where the engine offers one) the analytic gradient against the
oracle's.  Derandomised (the same examples every run).  The same property runs on the GPU as
tests/test_gpu_parity.py::test_random_problems_match_the_oracle with sizes a GPU likes."""

import os
import sys

import pytest
from hypothesis import HealthCheck, given, settings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")


@pytest.fixture(scope="module")
def emu():
    from inject import installed

    with installed() as lib:
        yield lib


from problem_gen import check_problem, problems  # noqa: E402


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(problems())
def test_random_problems_match_the_oracle_on_the_cpu_build(emu, prob):
    check_problem(*prob, grad=True)


def test_random_sharded_problems_match_the_oracle_on_the_cpu_build():
    """the same property for the SHARDED driver's device path (one rank, DeviceBackend): tests/emu/sharded_fuzz.py,
    derandomised, in a process of its own (torch's CUDA entry points are faked there)"""
    import subprocess

    # HIPEMU_POISON=ff: every never-written double of the "device" memory (hipMalloc of the CPU build, torch.empty of the
    # driver) is a NaN instead of whatever fresh pages hold - a don't-care entry that reached a result would show
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "sharded_fuzz.py"), "40", "330", "derandomize"],
                       env=dict(os.environ, HIPEMU_POISON="ff"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok: 40 sharded problems" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_random_problems_with_nan_poisoned_device_memory():
    """the engine's property under HIPEMU_POISON=ff (fresh device memory = NaN): the strict upper triangles, padding rows
    and workspace tails the kernels treat as don't-care must never contaminate a result"""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "fuzz_campaign.py"), "60", "200", "20260930"],
                       env=dict(os.environ, HIPEMU_POISON="ff"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok: 60 problems" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
