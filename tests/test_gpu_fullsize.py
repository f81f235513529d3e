"""GPU, large sizes: the HIP path against the REFERENCE BUILD (oracle/_ref, the reference's own scan_avx_4)
on a 10^8-code list, and size-independent properties on the 10^9-code list of bench.py
(sampled per-code values, shard/merge consistency, determinism)."""
import numpy as np
import pytest

from helpers import float_tables, heaps_equal

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


def test_1e8_codes_against_reference_build(pyqadc, po):
    if not po.have_ref():
        pytest.skip("oracle/_ref not present")
    M, n, R, keep = 16, 100_000_000, 100, 0.01
    idx = pyqadc.Index(M)
    idx.add_partition_synthetic(n, SEED)
    idx.finalize(keep)
    rng = np.random.default_rng(1)
    nq = 11                                              # two multi-query groups (8 + 3) launched as L2-sharing siblings
    tables = float_tables(rng, nq, 1, M)
    res = idx.query_scan(np.zeros((nq, 1), np.int32), tables.copy(), R, want_qtables=True)
    codes = po.fill_codes(0, n, SEED).reshape(n, 8)
    inter = po.ref_interleave(codes)
    for q in range(nq):
        # the reference's own AVX2 kernel on the same 8-byte codes and the int8 tables the device produced
        want = po.ref_scan_interleaved(M, [inter], [n], None, res["qtables"][q], R)
        assert heaps_equal(res["heaps"][q], want), q
    # and the float stages against the oracle restatement for one query (10^6 starts)
    w = po.query_scan(M, [codes], None, keep, [0], tables[0].copy(), R)
    assert res["qmax"][0] == np.float32(w["qmax"]) and np.array_equal(res["qtables"][0], w["qtables"])
    idx.close()


def test_1e9_codes_properties(pyqadc, po):
    from pyqadc import sharded
    M, n, R, keep = 16, 1_000_000_000, 100, 0.01
    rng = np.random.default_rng(2)
    tables = float_tables(rng, 2, 1, M)
    assign = np.zeros((2, 1), np.int32)
    whole = pyqadc.Index(M)
    whole.add_partition_synthetic(n, SEED)
    whole.finalize(keep)
    a = whole.query_scan(assign, tables.copy(), R, want_qtables=True)
    b = whole.query_scan(assign, tables.copy(), R)
    for q in range(2):
        assert heaps_equal(a["heaps"][q], b["heaps"][q])                      # deterministic
        assert len(a["heaps"][q][0]) == R and a["heaps"][q][1].max() == a["heaps"][q][1][0]   # full max-heap
    # per-code int8 sums on sampled windows == CPU oracle on the same generator stream
    cand = whole.candidates_i8(0, a["qtables"][0, 0])
    for first in (0, 123_456_784, 999_990_000):
        cnt = 10_000
        codes = po.fill_codes(first, cnt, SEED).reshape(cnt, 8)
        assert np.array_equal(cand[first:first + cnt], po.candidates_i8(M, codes, a["qtables"][0, 0]))
    # every returned neighbour carries its own int8 sum, and nothing below the heap's maximum was missed
    for q in range(2):
        cq = whole.candidates_i8(0, a["qtables"][q, 0]) if q else cand
        keys, vals = a["heaps"][q]
        assert np.array_equal(cq[keys], vals)
        assert int((cq < vals.max()).sum()) <= R
        assert set(np.flatnonzero(cq < vals.max()).tolist()) <= set(keys.tolist())
    whole.close()
    del cand
    # the same list cut into 4 shards (as 4 ranks would hold it): merged streams replay to the same heaps
    streams = []
    starts = po.start_size(n, keep)
    for first, ln in sharded.shard_ranges(n, 4):
        s = pyqadc.Index(M)
        s.add_partition_synthetic_shard(n, first, ln, SEED, starts)
        s.finalize(keep)
        streams.append(s.query_scan_candidates(assign, tables.copy(), R)["streams"])
        s.close()
    for q in range(2):
        ks = np.concatenate([st[q][0] for st in streams])
        vs = np.concatenate([st[q][1] for st in streams])
        assert heaps_equal(pyqadc.replay_i8(ks, vs, R, sentinel=True), a["heaps"][q]), q
