import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "quick-adc_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_runtime_first():
    """Some GPU tests hand torch tensors to the library.  torch bundles its own HIP runtime; if the library's runtime
    (/opt/rocm) comes up first in the process, torch's later initialisation finds no device.  Bring torch's up first
    whenever a GPU is there, so that the order of the test files does not matter."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:                                  # no torch / no GPU: the CPU suite does not need it
        pass
    yield


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure only)."""
    import pyoracle
    pyoracle.build()
    return pyoracle
