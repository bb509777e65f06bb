import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's CPU kernels sum in a thread-count dependent order; the golden vectors were generated with 8 threads
    # (oracle/gen_golden*.py), so the oracle pins run with 8 wherever the suite runs (a 256-thread host moved three of them
    # by 1e-6 .. 3e-6 past their bounds).  Tests that time or need more threads set their own count.
    import torch
    torch.set_num_threads(8)


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _library_present():
    """The C-ABI library is built in-tree and git-ignored: a fresh checkout has none.  Build it once if it is MISSING
    (hipcc cross-compiles without a GPU); an existing library is used as it is."""
    from densematchingbenchmark_amd import build
    if not os.path.exists(build.LIB_PATH):
        build.build_library(verbose=False)
    if not os.path.exists(build.SHIM_PATH):     # the thin torch extension over the same C ABI (host code only)
        build.build_torch_shim(verbose=False)


@pytest.fixture
def single_chain():
    """The convolution launches of this test stay on the single-chain kernels whatever their size (ops.set_split_k(False) =
    DMB_CONV_SINGLE_CHAIN, include/dmb_hip.h): the tests that compare two forms of those kernels BIT for bit, or a batch with its
    single pairs, need launches whose summation order does not depend on the launch's size."""
    from densematchingbenchmark_amd import ops
    before = ops.split_k()
    ops.set_split_k(False)
    yield
    ops.set_split_k(before)
