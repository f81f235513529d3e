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
    # the gather also carries the next batch's sharded pre-scan values: rank r contributes extra[r]
    extra = (np.arange(3 * 7, dtype=np.float32).reshape(3, 7) + 100 * rank)
    K, V, S, G = sharded.merge_batch(local, nq, R, None, "cpu", cap=64, extra=extra)   # tiny cap: exercises the regrow round
    want_g = np.concatenate([np.arange(3 * 7, dtype=np.float32).reshape(3, 7) + 100 * r for r in range(world)], axis=1)
    assert np.array_equal(G, want_g)
    assert np.array_equal(sharded.gather_prescan(extra, "cpu"), want_g)
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


def _ivf_worker(rank, world, port, M, sizes, ma, R, tmax, seed, out_dir):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    import pyoracle as po
    from pyqadc import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    parts = [rng.integers(0, 256, (s, M // 2), dtype=np.uint8) for s in sizes]
    perm = rng.permutation(sum(sizes)).astype(np.uint32)
    labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
    nq = 4
    assign = np.stack([rng.choice(len(sizes), ma, replace=False) for _ in range(nq)])
    qt = rng.integers(0, tmax + 1, (nq, ma, M, 16)).astype(np.int8)
    ranges = [sharded.shard_ranges(s, world)[rank] for s in sizes]
    keys, vals, slots, offs = [], [], [], [0]
    for q in range(nq):
        ps = [parts[p][ranges[p][0]:ranges[p][0] + ranges[p][1]] for p in assign[q]]
        ls = [labels[p][ranges[p][0]:ranges[p][0] + ranges[p][1]] for p in assign[q]]
        k, v, sl = po.shards_stream(M, ps, ls, [sizes[p] for p in assign[q]], [ranges[p][0] for p in assign[q]], qt[q], R)
        keys.append(k); vals.append(v); slots.append(sl); offs.append(offs[-1] + len(k))
    local = dict(keys=np.concatenate(keys), vals=np.concatenate(vals), slots=np.concatenate(slots),
                 offsets=np.array(offs, np.int64))
    K, V, S = sharded.merge_batch(local, nq, R, None, "cpu", cap=1 << 12, ma=ma)
    for q in range(nq):
        wk, wv = po.scan_i8(M, [parts[p] for p in assign[q]], [labels[p] for p in assign[q]], qt[q], R)
        assert S[q] == len(wk) and np.array_equal(K[q, :S[q]], wk) and np.array_equal(V[q, :S[q]], wv), q
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("M,tmax", [(16, 3), (32, 12)])
def test_two_rank_ivf_merge_interleaves_by_assign_slot(tmp_path, po, M, tmax):
    """IVF with every partition range-sharded over 2 ranks: streams are interleaved (slot, rank, position)."""
    sizes = [5000, 37, 1200, 64, 3001, 16]
    port = 29500 + (os.getpid() + M) % 2000
    mp.spawn(_ivf_worker, args=(2, port, M, sizes, 4, 100, tmax, 99 + M, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
