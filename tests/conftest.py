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


def pytest_generate_tests(metafunc):
    """Every GPU parity test runs through the scan paths of the library: the level-structured one (QADC_WGQ=0:
    bound levels, many workgroups per run, candidate sort), the same with its first levels replaced by a head launch
    (QADC_HEAD_LEVEL=2), and the one-workgroup-per-query one (QADC_WGQ=2, wherever it is structurally possible).  The
    environment variables are read by qadc_index_create; a test that sets the `wgq` option itself overrides it."""
    if metafunc.module.__name__.endswith("test_bench_launch"):
        return                                             # bench.py picks its own paths (and clears QADC_WGQ)
    if getattr(metafunc.function, "_path_independent", False):
        return                                             # (stateless build entry points: no query path involved)
    if metafunc.definition.get_closest_marker("gpu") and "scan_path" in metafunc.fixturenames:
        metafunc.parametrize("scan_path", ["levels", "levels_head", "wgq"], indirect=True)


@pytest.fixture(autouse=True)
def scan_path(request, monkeypatch):
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("QADC_TEST_HOOKS", "1")         # (the library ignores the QADC_* hooks without it)
        monkeypatch.setenv("QADC_WGQ", "2" if mode.startswith("wgq") else "0")
        # "levels_head": the first two bound levels of every query are scanned by one head launch of the query kernel
        monkeypatch.setenv("QADC_HEAD_LEVEL", "2" if mode == "levels_head" else "0")
    return mode
