"""CPU: the C-ABI library loads, exports every symbol include/qadc.h declares, its host-side heap
replay is bit-exact against the reference golden vectors, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    if not os.path.exists(pyqadc.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return pyqadc


def test_exports_every_declared_symbol(pyqadc):
    hdr = open(os.path.join(ROOT, "include", "qadc.h")).read()
    declared = set(re.findall(r"\b(qadc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pyqadc.SYMBOLS), declared ^ set(pyqadc.SYMBOLS)
    lib = pyqadc.lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.qadc_version()


def test_the_option_list_is_thirty_names_and_the_tests_draw_them(pyqadc):
    """qadc_option_names() = the options qadc_set_option accepts: at most 30, each documented in include/qadc.h, and each drawn
    by the randomised parity sweep (tests/test_gpu_fuzz.py) — except `profile` (diagnostics), `wgq` / `head_level` (the scan_path
    parametrisation of tests/conftest.py), `table_form` (test_search_with_device_side_feeders) and the dist_* ones
    (tests/dist_worker.py, tests/dist_cases.py)."""
    names = pyqadc.option_names()
    assert len(names) == len(set(names)) <= 30
    hdr = open(os.path.join(ROOT, "include", "qadc.h")).read()
    fuzz = open(os.path.join(ROOT, "tests", "test_gpu_fuzz.py")).read()
    dist = open(os.path.join(ROOT, "tests", "dist_worker.py")).read() + open(os.path.join(ROOT, "tests", "dist_cases.py")).read()
    src = open(os.path.join(ROOT, "quick-adc_amd", "csrc", "qadc_capi.cpp")).read()
    assert sorted(set(re.findall(r'n == "([a-z_0-9]+)"', src))) == sorted(names)          # the setter's chain and the list agree
    elsewhere = {"profile", "wgq", "head_level", "table_form"}
    for n in names:
        assert '"%s"' % n in hdr, n
        if n.startswith("dist_"):
            assert n in dist, n
        elif n not in elsewhere:
            assert re.search(r"\b%s=" % n, fuzz), n


def test_host_replay_matches_reference_golden(pyqadc):
    g = golden_cases.load()
    for i in range(int(g["n_heap_cases"])):
        R = int(g["h%d_R" % i])
        k, v = pyqadc.replay_i8(g["h%d_in_keys" % i], g["h%d_in_vals" % i], R)
        assert np.array_equal(k, g["h%d_keys" % i]) and np.array_equal(v, g["h%d_vals" % i]), i


def test_sort_keys_matches_reference_golden(pyqadc, po):
    """kv_heap::sort_keys on the reference's final heap arrays == the reference's own bh.sort_keys() output
    (binheap.hpp:129-137; tie order of an unstable std::sort), all golden scan cases."""
    g = golden_cases.load()
    n = 0
    for c in golden_cases.scan_cases(g, po):
        assert np.array_equal(pyqadc.sort_keys_i8(c["keys"], c["vals"]), c["sorted"]), c["cid"]
        n += 1
    assert n >= 50


def test_host_replay_random_vs_oracle(pyqadc, po):
    rng = np.random.default_rng(0)
    for n, R, vmax in ((0, 5, 3), (1, 1, 3), (1000, 7, 2), (20000, 100, 126)):
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        vals = rng.integers(0, vmax + 1, n).astype(np.int8)
        for sentinel in (False, True):
            got = pyqadc.replay_i8(keys, vals, R, sentinel)
            kk = np.concatenate([[0], keys]).astype(np.uint32) if sentinel else keys
            vv = np.concatenate([[127], vals]).astype(np.int8) if sentinel else vals
            want = po.heap_replay_i8(kk, vv, R)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_no_gpu_fails_loudly(pyqadc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pyqadc.QadcError):
        pyqadc.Index(16)


def test_unsupported_m_rejected(pyqadc):
    h = C.c_void_p()
    rc = pyqadc.lib().qadc_index_create(C.byref(h), 8, 0)
    assert rc == -1 and b"Supported configurations are: (16,4) (32,4)" in pyqadc.lib().qadc_last_error()


def test_worker_pool_under_thread_sanitizer(tmp_path):
    """host/worker_pool.hpp (the host replay's persistent threads): 4000 jobs of 1..37 tasks on 1..9 threads, built with
    ThreadSanitizer — every task runs exactly once, results are visible after run(), no race report."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pool_tsan")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wall", "-Werror",
                           os.path.join(root, "tests", "cpp", "pool_test.cpp"), "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"pool ok" in p.stdout and b"ThreadSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
