import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle legs (torch / oneDNN on the host cores) with one thread per hardware thread of a 256-thread host thrash for minutes,
    more so when other tenants share the host (the GPU suite was seen at 13 minutes instead of 4): 16 threads are enough for the test sizes."""
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _no_miopen_for_torch_convs():
    """The few torch.nn.Conv2d ops the tests run through PyTorch itself (the `net` wrapped by the MVF module in
    test_mvf_gpu.py) go through ATen's native im2col + rocBLAS path instead of MIOpen: MIOpen JIT-compiles its kernels on
    first use on a fresh box and was seen to abort() the interpreter there once in ~10 runs.  Nothing under test uses MIOpen."""
    try:
        import torch
        torch.backends.cudnn.enabled = False
    except Exception:
        pass
    yield
