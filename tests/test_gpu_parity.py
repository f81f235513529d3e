"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact bar: heap arrays (keys, values, size), int8 tables, qmin/qmax."""
import numpy as np
import pytest

from helpers import rand_codes, rand_qtables, float_tables, heaps_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


@pytest.mark.parametrize("M", [16, 32])
@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 37, 1000, 4097, 100003])
def test_scan_i8_flat_matches_oracle(pyqadc, po, M, n):
    rng = np.random.default_rng(1000 * M + n)
    codes = rand_codes(rng, n, M)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(0.01)
    for tmax, R in [(3, 10), (12, 100), (127, 100), (40, 1)]:
        qt = rand_qtables(rng, (3, 1), M, tmax)
        got = idx.scan_i8(np.zeros((3, 1), np.int32), qt, R)
        for q in range(3):
            want = po.scan_i8(M, [codes], None, qt[q], R)
            assert heaps_equal(got[q], want), (M, n, tmax, R, q)
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
def test_candidates_all_codes(pyqadc, po, M):
    rng = np.random.default_rng(7 + M)
    n = 300001
    codes = rand_codes(rng, n, M)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(0.01)
    for tmax in (5, 127):
        qt = rand_qtables(rng, (), M, tmax)
        assert np.array_equal(idx.candidates_i8(0, qt), po.candidates_i8(M, codes, qt))
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
def test_scan_i8_ivf_labels_shared_heap(pyqadc, po, M):
    rng = np.random.default_rng(99 + M)
    sizes = [5000, 37, 0, 16, 1, 12345, 333, 70001]
    parts = [rand_codes(rng, s, M) for s in sizes]
    perm = rng.permutation(sum(sizes)).astype(np.uint32)
    labels, o = [], 0
    for s in sizes:
        labels.append(perm[o:o + s].copy())
        o += s
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.01)
    nq, ma = 6, 5
    assign = np.stack([rng.choice(len(sizes), ma, replace=False) for _ in range(nq)]).astype(np.int32)
    for tmax, R in [(6, 100), (30, 50)]:
        qt = rand_qtables(rng, (nq, ma), M, tmax)
        got = idx.scan_i8(assign, qt, R)
        for q in range(nq):
            ps = [parts[p] for p in assign[q]]
            ls = [labels[p] for p in assign[q]]
            want = po.scan_i8(M, ps, ls, qt[q], R)
            assert heaps_equal(got[q], want), (M, tmax, R, q)
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
@pytest.mark.parametrize("levels", [(16, 2), (1024, 16)])
def test_query_scan_full_pipeline(pyqadc, po, M, levels):
    rng = np.random.default_rng(5 + M)
    sizes = [40000, 25000, 16, 9000, 100001]
    parts = [rand_codes(rng, s, M) for s in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, [np.arange(s, dtype=np.uint32) * 3 + 1 for s in sizes])
    labels = [np.arange(s, dtype=np.uint32) * 3 + 1 for s in sizes]
    keep = 0.02
    idx.finalize(keep)
    idx.set_option("level_base", levels[0])
    idx.set_option("level_growth", levels[1])
    nq, ma, R = 5, 3, 100
    assign = np.stack([rng.choice(len(sizes), ma, replace=False) for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M, negatives=True)
    t_gpu = tables.copy()
    res = idx.query_scan(assign, t_gpu, R, want_qtables=True)
    for q in range(nq):
        t_cpu = tables[q].copy()
        want = po.query_scan(M, parts, labels, keep, assign[q], t_cpu, R, quant_mode=1)
        assert want["rc"] == 0 and res["status"][q] == 0
        assert np.float32(want["qmin"]) == res["qmin"][q] and np.float32(want["qmax"]) == res["qmax"][q]
        assert np.array_equal(want["qtables"], res["qtables"][q])
        assert np.array_equal(t_cpu, t_gpu[q])          # in-place clamp of negatives
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    idx.close()
