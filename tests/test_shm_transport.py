"""CPU: the shared-memory all-gather behind qadc_dist_init_transport (qadc_shm_transport_*), host-buffer form — several
PROCESSES exchange blocks of changing sizes through one segment; a missing rank turns into an error, not a hang; the
size-balanced placement helper.  (The device form is exercised by tests/test_gpu_dist_multiproc.py.)"""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import pyqadc
rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
t = pyqadc.ShmTransport(name, rank, world, slot_bytes=1 << 20, timeout_s=float(sys.argv[4]))
for it, n in enumerate([1, 7, 4096, 100000, 3, 131072]):
    mine = (np.arange(n, dtype=np.uint64) * 1000003 + rank * 7919 + it).astype(np.uint64)
    got = t.allgather_host(mine)
    for g in range(world):
        want = (np.arange(n, dtype=np.uint64) * 1000003 + g * 7919 + it).astype(np.uint64)
        assert np.array_equal(got[g], want), (it, g)
try:
    t.allgather_host(np.zeros((1 << 20) + 8, np.uint8))          # beyond slot_bytes: an error on every rank alike
    raise SystemExit(5)
except pyqadc.QadcError:
    pass
t.close()
print("ok")
""" % os.path.join(ROOT, "quick-adc_amd")


def _spawn(rank, world, name, timeout_s):
    return subprocess.Popen([sys.executable, "-c", CHILD, str(rank), str(world), name, str(timeout_s)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.parametrize("world", [1, 2, 5])
def test_shm_allgather_between_processes(world):
    name = "/qadc_cpu_%s" % uuid.uuid4().hex[:12]
    procs = [_spawn(r, world, name, 60) for r in range(world)]
    try:
        for r, p in enumerate(procs):
            so, se = p.communicate(timeout=120)
            assert p.returncode == 0 and so.strip() == b"ok", (r, se.decode()[-800:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert not os.path.exists("/dev/shm" + name)              # rank 0 unlinked the segment


def test_a_missing_rank_is_an_error_not_a_hang():
    name = "/qadc_cpu_%s" % uuid.uuid4().hex[:12]
    p = _spawn(0, 2, name, 1.5)                               # rank 1 never shows up
    try:
        so, se = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode != 0 and b"timed out" in se


def test_place_partitions_is_size_balanced_and_deterministic():
    import pyqadc
    rng = np.random.default_rng(3)
    sizes = rng.multinomial(10 ** 7, np.ones(4096) / 4096).astype(np.uint32)
    owner = pyqadc.place_partitions(sizes, 8)
    load = np.bincount(owner, weights=sizes, minlength=8)
    assert load.max() - load.min() <= sizes.max()             # greedy longest-first: within one partition of each other
    assert np.array_equal(owner, pyqadc.place_partitions(sizes, 8))
    assert list(pyqadc.place_partitions([5, 9, 1, 7, 3, 3], 3)) == [2, 0, 2, 1, 2, 1]
