import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption(
        "--emu", action="store_true",
        help="development aid while no GPU is at hand: run (a selection of) the -m gpu tests against the CPU build of the "
             "kernel sources (tests/emu); tests that need torch CUDA tensors still need the GPU",
    )


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--emu"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from inject import fake_cuda_tensors, installed

        fake_cuda_tensors()
        config._emu = installed()
        config._emu.__enter__()


# The driver runs the GPU suite with -x: what is most basic (and longest validated on the hardware) goes first, so that a
# failure in a younger layer (adaptor, sharded driver, optional schedules) can not cut the core parity record short.
GPU_ORDER = ["test_gpu_parity", "test_gpu_pins_and_sizes", "test_gpu_slab_layout", "test_gpu_adaptor", "test_gpu_sharded", "test_gpu_c_caller",
             "test_gpu_zz_optional_schedules"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return GPU_ORDER.index(mod) if mod in GPU_ORDER else -1  # CPU modules keep their place in front

    items.sort(key=key)  # stable: the order inside a module is untouched


def pytest_unconfigure(config):
    if getattr(config, "_emu", None) is not None:
        config._emu.__exit__(None, None, None)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
