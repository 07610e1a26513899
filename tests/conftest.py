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


def pytest_unconfigure(config):
    if getattr(config, "_emu", None) is not None:
        config._emu.__exit__(None, None, None)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
