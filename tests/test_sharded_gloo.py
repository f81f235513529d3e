"""CPU, world_size 2 over gloo: the cross-rank merge of per-shard push streams (pyqadc/sharded.py)
reproduces the single sequential scan of the whole list, heap array for heap array — including
tie-heavy tables where gathering per-shard top-R would NOT be exact."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, M, n, R, tmax, seed, out_dir):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (sys.path setup)
    import pyoracle as po
    from pyqadc import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)                    # same data on every rank
    codes = rng.integers(0, 256, (n, M // 2), dtype=np.uint8)
    nq = 5
    qt = rng.integers(0, tmax + 1, (nq, M, 16)).astype(np.int8)
    first, ln = sharded.shard_ranges(n, world)[rank]
    keys, vals, offs = [], [], [0]
    for q in range(nq):
        k, v = po.shard_stream(M, codes[first:first + ln], None, n, first, qt[q], R)
        keys.append(k)
        vals.append(v)
        offs.append(offs[-1] + len(k))
    local = dict(keys=np.concatenate(keys), vals=np.concatenate(vals), offsets=np.array(offs, np.int64))
    K, V, S = sharded.merge_batch(local, nq, R, None, "cpu", cap=64)   # tiny cap: exercises the regrow round
    for q in range(nq):
        wk, wv = po.scan_i8(M, [codes], None, qt[q:q + 1], R)
        assert S[q] == len(wk), (q, S[q], len(wk))
        assert np.array_equal(K[q, :S[q]], wk) and np.array_equal(V[q, :S[q]], wv), q
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("M,n,R,tmax", [(16, 5003, 100, 2), (16, 20000, 10, 30), (32, 777, 100, 6)])
def test_two_rank_merge_equals_sequential_scan(tmp_path, po, M, n, R, tmax):
    port = 29500 + (os.getpid() + n) % 2000
    mp.spawn(_worker, args=(2, port, M, n, R, tmax, 1234 + n, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_shard_ranges_block_aligned():
    from pyqadc import sharded
    for n, w in ((1000000000, 8), (100003, 2), (37, 2), (16, 4)):
        rs = sharded.shard_ranges(n, w)
        assert sum(l for _, l in rs) == n and rs[0][0] == 0
        for (f, l), (f2, _) in zip(rs, rs[1:]):
            assert f + l == f2 and f % 16 == 0 and l % 16 == 0
