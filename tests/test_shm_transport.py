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


MISMATCH = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import pyqadc
rank, name = int(sys.argv[1]), sys.argv[2]
t = pyqadc.ShmTransport(name, rank, 2, slot_bytes=1 << 16, timeout_s=30)
t.allgather_host(np.zeros(64, np.uint8))
try:
    t.allgather_host(np.zeros(64 if rank == 0 else 128, np.uint8))   # the ranks are in DIFFERENT collectives
    raise SystemExit(5)
except pyqadc.QadcError as e:
    assert "collective mismatch" in str(e) or "aborted" in str(e), str(e)
    print("mismatch" if "collective mismatch" in str(e) else "aborted")
""" % os.path.join(ROOT, "quick-adc_amd")


def test_ranks_in_different_collectives_fail_instead_of_exchanging_garbage():
    """the transport used to copy bytes_per_rank out of every slot whatever the other ranks had put there."""
    name = "/qadc_cpu_%s" % uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, "-c", MISMATCH, str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    outs = []
    try:
        for r, p in enumerate(procs):
            so, se = p.communicate(timeout=120)
            assert p.returncode == 0, (r, se.decode()[-800:])
            outs.append(so.strip())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)
    assert b"mismatch" in outs                                  # at least one rank names the mismatch; the other may see the poison first


def test_a_stale_segment_of_a_crashed_run_is_not_joined():
    """a segment left by a crashed run (full size, valid magic, same geometry, abort flag set) under the same
    name: rank 1 arrives FIRST, must not join it, and meets rank 0 on the segment rank 0 creates afterwards."""
    import struct
    import time
    name = "/qadc_cpu_%s" % uuid.uuid4().hex[:12]
    world, slot = 2, 1 << 20
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    with open("/dev/shm" + name, "wb") as f:                   # ShmHeader: magic, arrived, generation, abort, world, creator_pid, slot_bytes
        f.write(struct.pack("<6IQ", 0x51414443, 1, 7, 1, world, dead.pid, slot))
        f.truncate(4096 + world * slot)
    p1 = _spawn(1, world, name, 60)
    time.sleep(1.0)
    p0 = _spawn(0, world, name, 60)
    try:
        for r, p in ((0, p0), (1, p1)):
            so, se = p.communicate(timeout=120)
            assert p.returncode == 0 and so.strip() == b"ok", (r, se.decode()[-800:])
    finally:
        for p in (p0, p1):
            if p.poll() is None:
                p.kill()
        if os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)
