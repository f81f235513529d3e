"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact bar: heap arrays (keys, values, size), int8 tables, qmin/qmax."""
import numpy as np
import pytest

from helpers import rand_codes, rand_qtables, float_tables, heaps_equal, path_independent

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


def test_gpu_matches_reference_golden(pyqadc, po):
    """HIP path vs the vectors the reference's own scan_avx_4 produced (tests/golden)."""
    import golden_cases
    g = golden_cases.load()
    n = 0
    for c in golden_cases.scan_cases(g, po):
        idx = pyqadc.Index(c["M"])
        idx.add_partitions(c["parts"], c["labels"])
        idx.finalize(0.01)
        nparts = len(c["parts"])
        got = idx.scan_i8(np.arange(nparts, dtype=np.int32).reshape(1, nparts), c["qt"], c["R"])[0]
        assert np.array_equal(got[0], c["keys"]) and np.array_equal(got[1], c["vals"]), c["cid"]
        idx.close()
        n += 1
    assert n >= 50


@pytest.mark.parametrize("M", [16, 32])
def test_import_of_reference_block_layout(pyqadc, po, M):
    rng = np.random.default_rng(31 + M)
    for n in (1, 16, 37, 5000):
        codes = rand_codes(rng, n, M)
        idx = pyqadc.Index(M)
        idx.add_partition_interleaved(po.interleave(codes), n)
        idx.finalize(0.01)
        assert np.array_equal(idx.read_codes(0, 0, n), codes)
        qt = rand_qtables(rng, (1, 1), M, 9)
        assert heaps_equal(idx.scan_i8(np.zeros((1, 1), np.int32), qt, 100)[0], po.scan_i8(M, [codes], None, qt[0], 100))
        idx.close()


def test_synthetic_generator_matches_cpu(pyqadc, po):
    idx = pyqadc.Index(16)
    idx.add_partition_synthetic(100001, 0xABCDEF)
    idx.add_partition_synthetic_shard(100000, 4096, 5000, 77, 1000)
    idx.finalize(0.01)
    assert np.array_equal(idx.read_codes(0, 0, 100001).reshape(-1), po.fill_codes(0, 100001, 0xABCDEF))
    assert np.array_equal(idx.read_codes(1, 0, 5000).reshape(-1), po.fill_codes(4096, 5000, 77))
    idx.close()


@pytest.mark.parametrize("M,n,tmax", [(16, 100003, 3), (16, 300000, 25), (32, 50001, 8)])
def test_sharded_streams_replay_to_sequential_heap(pyqadc, po, M, n, tmax):
    """Two shards of one list on one GPU (the per-rank work of the multi-GPU path): concatenating the
    shard streams in shard order and replaying equals the single sequential scan; and the full
    float pipeline gives every shard the same qmax / int8 tables."""
    from pyqadc import sharded
    rng = np.random.default_rng(n)
    codes = rand_codes(rng, n, M)
    keep, R, nq = 0.01, 100, 4
    starts_n = po.start_size(n, keep)
    shards = []
    for first, ln in sharded.shard_ranges(n, 3):
        idx = pyqadc.Index(M)
        idx.add_partition_shard(codes[first:first + ln], first, n, starts=codes[:starts_n])
        idx.finalize(keep)
        shards.append(idx)
    tables = float_tables(rng, nq, 1, M)
    assign = np.zeros((nq, 1), np.int32)
    res = [s.query_scan_candidates(assign, tables.copy(), R) for s in shards]
    for q in range(nq):
        want = po.query_scan(M, [codes], None, keep, [0], tables[q].copy(), R)
        for r in res:
            assert r["qmax"][q] == np.float32(want["qmax"]) and r["qmin"][q] == np.float32(want["qmin"])
        ks = np.concatenate([r["streams"][q][0] for r in res])
        vs = np.concatenate([r["streams"][q][1] for r in res])
        assert heaps_equal(pyqadc.replay_i8(ks, vs, R, sentinel=True), (want["keys"], want["values"])), q
    for s in shards:
        s.close()


def test_qmax_too_high_is_reported_not_fatal(pyqadc, po):
    rng = np.random.default_rng(4)
    codes = rand_codes(rng, 5000, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.001)                      # 5 starts < R-1: the reference would exit(1)
    tables = float_tables(rng, 2, 1, 16)
    res = idx.query_scan(np.zeros((2, 1), np.int32), tables, 100)
    assert list(res["status"]) == [1, 1] and list(res["sizes"]) == [0, 0] and res["qmax"][0] > 1e30
    idx.close()


def test_candidate_buffer_regrow_is_exact(pyqadc, po):
    rng = np.random.default_rng(8)
    codes = rand_codes(rng, 200000, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.01)
    idx.set_option("wgq", 0)                # this test is about the level-structured path's own fallbacks
    idx.set_option("head_level", 0)         # (with every bound level launched separately)
    idx.set_option("cand_capacity", 16)      # far too small: forces the overflow -> regrow -> rerun path
    idx.set_option("profile", 1)
    qt = rand_qtables(rng, (2, 1), 16, 20)
    got = idx.scan_i8(np.zeros((2, 1), np.int32), qt, 100)
    assert idx.profile()["regrows"] >= 1
    for q in range(2):
        assert heaps_equal(got[q], po.scan_i8(16, [codes], None, qt[q], 100))
    idx.close()


def test_negative_int8_tables_rejected(pyqadc):
    idx = pyqadc.Index(16)
    idx.add_partitions([np.zeros((10, 8), np.uint8)])
    idx.finalize(0.01)
    qt = np.zeros((1, 1, 16, 16), np.int8)
    qt[0, 0, 3, 3] = -1
    with pytest.raises(pyqadc.QadcError):
        idx.scan_i8(np.zeros((1, 1), np.int32), qt, 10)
    idx.close()


@pytest.mark.parametrize("variant", [0x00, 0x04, 0x08, 0x0c])
@pytest.mark.parametrize("M", [16, 32])
def test_kernel_tuning_variants_are_exact(pyqadc, po, M, variant):
    """Every form of the scan kernel (non-temporal loads, chunked tile order, both, neither) must give the same heap arrays."""
    rng = np.random.default_rng(variant * 7 + M)
    sizes = [70001, 4097, 1]
    parts = [rand_codes(rng, s, M) for s in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(0.01)
    idx.set_option("variant", variant)
    idx.set_option("level_base", 64)
    idx.set_option("level_growth", 4)
    for wgs in (0, 3):
        idx.set_option("wgs_per_item", wgs)
        qt = rand_qtables(rng, (2, 3), M, 11)
        got = idx.scan_i8(np.array([[0, 1, 2], [2, 0, 1]], np.int32), qt, 100)
        for q, order in enumerate(([0, 1, 2], [2, 0, 1])):
            want = po.scan_i8(M, [parts[p] for p in order], None, qt[q], 100)
            assert heaps_equal(got[q], want), (M, variant, wgs, q)
    idx.close()


def test_host_sort_fallback_when_a_query_has_too_many_candidates(pyqadc, po):
    """One query emits far more than the device sort handles (all-zero tables + a huge first level):
    its region is regrown and sorted on the host; the other query stays on the device-sorted path."""
    rng = np.random.default_rng(12)
    codes = rand_codes(rng, 150001, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.01)
    idx.set_option("wgq", 0)                # this test is about the level-structured path's own fallbacks
    idx.set_option("head_level", 0)         # (with every bound level launched separately)
    idx.set_option("level_base", 1 << 20)
    idx.set_option("profile", 1)
    qt = rand_qtables(rng, (2, 1), 16, 30)
    qt[0] = 0                                             # every code ties at 0 and is emitted by level 0
    got = idx.scan_i8(np.zeros((2, 1), np.int32), qt, 100)
    assert idx.profile()["regrows"] >= 1
    for q in range(2):
        assert heaps_equal(got[q], po.scan_i8(16, [codes], None, qt[q], 100)), q
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
def test_two_phase_prescan_is_exact(pyqadc, po, M):
    """Starts beyond the sample are filtered by the sample's R-th smallest distance before the select:
    qmax (and everything downstream) must not change."""
    rng = np.random.default_rng(21 + M)
    sizes = [300000, 150000, 90001]
    parts = [rand_codes(rng, s, M) for s in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    keep = 0.05                                            # 15000 + 7500 + 4500 starts
    idx.finalize(keep)
    idx.set_option("prescan_sample", 1000)
    nq, ma, R = 4, 3, 100
    assign = np.stack([rng.permutation(3) for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M)
    res = idx.query_scan(assign, tables.copy(), R, want_qtables=True)
    for q in range(nq):
        want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
        assert res["qmax"][q] == np.float32(want["qmax"]) and np.array_equal(res["qtables"][q], want["qtables"])
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"]))
    idx.close()


def test_prescan_survivor_overflow_falls_back_to_full_prescan(pyqadc, po):
    """Adversarial starts: everything after the sample beats the sample's R-th smallest, so the survivor
    buffer overflows; the batch is re-run with an unfiltered pre-scan and stays exact."""
    rng = np.random.default_rng(5)
    n, M, keep = 2000000, 16, 0.05
    codes = rand_codes(rng, n, M)
    codes[4096:100000] = 0                                # starts 4096.. all hit centroid 0 of every sub-quantizer
    tables = float_tables(rng, 2, 1, M) + np.float32(1.0)
    tables.reshape(2, M, 16)[:, :, 0] = np.float32(0.001)  # ... which is by far the closest
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(keep)
    idx.set_option("wgq", 0)                # this test is about the level-structured path's own fallbacks
    idx.set_option("head_level", 0)         # (with every bound level launched separately)
    idx.set_option("prescan_sample", 4096)
    idx.set_option("profile", 1)
    res = idx.query_scan(np.zeros((2, 1), np.int32), tables.copy(), 100, want_qtables=True)
    assert idx.profile()["regrows"] >= 1
    for q in range(2):
        want = po.query_scan(M, [codes], None, keep, [0], tables[q].copy(), 100)
        assert res["qmax"][q] == np.float32(want["qmax"])
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"]))
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
def test_edge_cases_small_r_large_r_repeated_and_empty_partitions(pyqadc, po, M):
    """R larger than the list (heap never fills, sentinel survives), R = 1, the same partition probed
    twice, and empty partitions inside assign[] (skipped like db_query_4.cpp:291-293)."""
    rng = np.random.default_rng(77 + M)
    sizes = [50, 0, 333, 0, 17]
    parts = [rand_codes(rng, s, M) for s in sizes]
    labels = [np.arange(s, dtype=np.uint32) + 1000 * i for i, s in enumerate(sizes)]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.5)
    assign = np.array([[0, 1, 2, 3], [2, 2, 4, 0], [4, 3, 1, 0]], np.int32)
    for R in (1, 7, 100, 1000):
        qt = rand_qtables(rng, (3, 4), M, 9)
        got = idx.scan_i8(assign, qt, R)
        for q in range(3):
            ps = [parts[p] for p in assign[q]]
            ls = [labels[p] for p in assign[q]]
            want = po.scan_i8(M, ps, ls, qt[q], R)
            assert heaps_equal(got[q], want), (M, R, q)
    # full float pipeline with starts taken from every probed partition (keep = 50 %)
    tables = float_tables(rng, 3, 4, M)
    res = idx.query_scan(assign, tables.copy(), 100)
    for q in range(3):
        want = po.query_scan(M, parts, labels, 0.5, assign[q], tables[q].copy(), 100)
        assert want["rc"] == res["status"][q]
        if want["rc"] == 0:
            assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


def _coarse_dists(x, c):
    """The coarse distances of the query row(s) x [1 or n][dim] to the centroids c [K][dim] as find_k_neighbors gets them
    (compute_cross_dists_blas, distances.hpp:151-183): the ORACLE's orc_cross_dists — (||x||^2 + ||c||^2) with the norms as the
    reference compiles them, then -2 x.c as one sequential dot.  Returns [K] for one row, [n][K] for several."""
    import pyoracle
    d = pyoracle.cross_dists(c, np.ascontiguousarray(x, np.float32).reshape(-1, c.shape[1]))
    return d[0] if d.shape[0] == 1 else d


def _seq_expansion(x, c):
    """The BLAS-expansion table form (distances.hpp:151-183, 277-292) of one vector (x [M][1][ds], c = codebooks
    [M][16][ds]) -> [M][16], from the ORACLE (orc_tables_expansion: (||x||^2 + ||c||^2) with the norms as the reference
    compiles them, then -2 x.c as one sequential dot): what the device feeder and host/query_driver.hpp must reproduce."""
    import pyoracle
    return pyoracle.tables_expansion(c, x.reshape(-1)).reshape(c.shape[0], 16)


@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("M,K,ma,opq", [(16, 37, 5, False), (32, 16, 3, False), (16, 0, 1, False), (16, 21, 4, True),
                                         (32, 0, 1, True)])
def test_search_with_device_side_feeders(pyqadc, po, M, K, ma, opq, form):
    """N1: queries in -> coarse assignment, residuals and float tables on the GPU -> same heaps as feeding
    the oracle with tables/assignments evaluated by the same float loops on the host."""
    rng = np.random.default_rng(M * 100 + K)
    dim, nq, R, keep = 128, 7, 100, 0.05
    ds = dim // M
    cb = rng.normal(size=(M, 16, ds)).astype(np.float32)
    nparts = max(K, 1)
    sizes = [int(s) for s in rng.integers(3000, 9000, nparts)]
    parts = [rand_codes(rng, s, M) for s in sizes]
    labels = None
    if K:
        perm = rng.permutation(sum(sizes)).astype(np.uint32)
        labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    idx.set_pq(cb)
    coarse = None
    if K:
        coarse = rng.normal(size=(K, dim)).astype(np.float32)
        coarse[5] = coarse[3]                                   # exact distance tie between two centroids
        idx.set_coarse(coarse)
    rot = None
    if opq:
        rot = np.linalg.qr(rng.normal(size=(dim, dim)))[0].astype(np.float32)
        idx.set_rotation(rot)
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    idx.set_option("table_form", form)      # 0 direct, 1 BLAS expansion, 2 (default) nns_engine's rule: expansion iff ma > 1
    # direct form: added like the reference's fmanorm as compiled (oracle's orc_tables_direct, pinned to the reference build)
    direct_fn = lambda r, cb_: po.tables_direct(cb_, r.reshape(-1)).reshape(M, 16)
    table_fn = _seq_expansion if (form == 1 or (form == 2 and ma > 1)) else direct_fn
    res = idx.search(queries, ma, R)
    for q in range(nq):
        if K:
            dist = _coarse_dists(queries[q][None, :], coarse)
            assign = po.select_k_neighbors(dist, ma)[0][0]       # find_k_neighbors' heaps (centroids 3 and 5 tie exactly)
            resid = (queries[q][None, :] - coarse[assign]).astype(np.float32)
        else:
            assign = np.zeros(ma, np.int32)
            resid = np.repeat(queries[q][None, :], ma, 0)
        assert np.array_equal(res["assign"][q], assign), q
        if opq:   # rotated[r] = sum_c x[c] * rot[r][c], float32, ascending c
            acc = np.zeros((ma, dim), np.float32)
            for cc in range(dim):
                acc = (acc + (resid[:, cc:cc + 1] * rot[None, :, cc]).astype(np.float32)).astype(np.float32)
            resid = acc
        tables = np.zeros((ma, M, 16), np.float32)
        for a in range(ma):
            tables[a] = table_fn(resid[a].reshape(M, 1, ds), cb)
        want = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(tables.reshape(ma, M * 16)), R)
        assert want["rc"] == res["status"][q] == 0
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


def test_collect_candidates_capacity_retry(pyqadc, po):
    rng = np.random.default_rng(3)
    codes = rand_codes(rng, 120000, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.01)
    tables = float_tables(rng, 3, 1, 16)
    idx.submit(0, np.zeros((3, 1), np.int32), tables.copy(), 100)
    res = idx.collect_candidates(0, capacity=8)            # far too small: E_CAPACITY, then the retry path
    for q in range(3):
        a, b = int(res["offsets"][q]), int(res["offsets"][q + 1])
        want = po.query_scan(16, [codes], None, 0.01, [0], tables[q].copy(), 100)
        assert heaps_equal(pyqadc.replay_i8(res["keys"][a:b], res["vals"][a:b], 100, sentinel=True),
                           (want["keys"], want["values"]))
    idx.close()


def test_sort_paths_wave_segments_and_workgroup_fallback(pyqadc, po):
    """Level segments up to 1024 candidates are ordered by one wave each; a bigger segment (here a 4096-code
    first level that emits everything) takes the workgroup-wide sort.  Both must give the oracle's heap."""
    rng = np.random.default_rng(19)
    parts = [rand_codes(rng, 30011, 16), rand_codes(rng, 9000, 16)]
    labels = [np.arange(30011, dtype=np.uint32)[::-1].copy(), np.arange(9000, dtype=np.uint32) + 50000]
    idx = pyqadc.Index(16)
    idx.add_partitions(parts, labels)
    idx.finalize(0.01)
    idx.set_option("profile", 1)
    for base in (512, 4096):
        idx.set_option("level_base", base)
        qt = rand_qtables(rng, (3, 2), 16, 5)              # tie-heavy; nearly every code of level 0 is emitted
        got = idx.scan_i8(np.array([[0, 1], [1, 0], [0, 1]], np.int32), qt, 100)
        for q, order in enumerate(([0, 1], [1, 0], [0, 1])):
            want = po.scan_i8(16, [parts[p] for p in order], [labels[p] for p in order], qt[q], 100)
            assert heaps_equal(got[q], want), (base, q)
    assert idx.profile()["host_sorted_queries"] == 0
    idx.close()


def test_large_heap_capacity(pyqadc, po):
    """R = 1000: level segments exceed what one wave orders and the total exceeds the device sort capacity for
    some queries (regrow + workgroup-wide / host ordering); results stay exact."""
    rng = np.random.default_rng(41)
    codes = rand_codes(rng, 400000, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.02)
    tables = float_tables(rng, 3, 1, 16)
    res = idx.query_scan(np.zeros((3, 1), np.int32), tables.copy(), 1000)
    for q in range(3):
        want = po.query_scan(16, [codes], None, 0.02, [0], tables[q].copy(), 1000)
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


def test_many_indexes_create_destroy(pyqadc, po):
    rng = np.random.default_rng(42)
    codes = rand_codes(rng, 5000, 16)
    qt = rand_qtables(rng, (1, 1), 16, 9)
    want = po.scan_i8(16, [codes], None, qt[0], 50)
    live = []
    for i in range(12):
        idx = pyqadc.Index(16)
        idx.add_partitions([codes])
        idx.finalize(0.01)
        assert heaps_equal(idx.scan_i8(np.zeros((1, 1), np.int32), qt, 50)[0], want)
        if i % 3 == 0:
            live.append(idx)            # several indexes alive at once on the same GPU
        else:
            idx.close()
    for idx in live:
        assert heaps_equal(idx.scan_i8(np.zeros((1, 1), np.int32), qt, 50)[0], want)
        idx.close()


@path_independent
def test_pq_encode_writes_the_reference_codes_golden(pyqadc, po):
    """N4: qadc_pq_encode / qadc_ivf_encode_host (OPQ cases) write the codes of tests/golden/ref_encode_cases.npz — made by the
    reference's own extract_subvectors, compute_cross_dists_blas (up to its sgemm), add_candidates_heaps and
    multiple_set_bits_4 (oracle/gen_golden_encode.py) — on normal, grid-valued (exact ties), duplicate-centroid,
    midpoint and cancellation-dominated inputs, M = 16 / 32, sq_dim 4 ... 60, with and without OPQ; encode_form 0 writes
    the oracle's direct-form codes, which differ from the reference's on the last two kinds."""
    import golden_cases
    ncase = differs = 0
    for c in golden_cases.encode_cases():
        cb, v, rot = c["codebooks"], c["vectors"], c["rotation"]
        if rot is None:
            got = pyqadc.pq_encode(cb, v)
            got0 = pyqadc.pq_encode(cb, v, encode_form=0)
        else:
            got = pyqadc.ivf_encode(cb, v, rotation=rot)[1]
            got0 = pyqadc.ivf_encode(cb, v, rotation=rot, encode_form=0)[1]
        assert np.array_equal(got, c["codes"]), c["cid"]
        assert np.array_equal(got0, po.pq_encode(cb, v, rot, form=0)), c["cid"]
        differs += int((got0 != got).any())
        ncase += 1
    assert ncase == 15 and differs >= 4


@path_independent
@pytest.mark.parametrize("M,dim", [(16, 128), (32, 128), (32, 96), (16, 64), (16, 480)])
@pytest.mark.parametrize("kind", ["normal", "grid", "offset", "mid"])
def test_pq_encode_matches_the_oracle_encoder(pyqadc, po, M, dim, kind):
    """N4 at larger counts: device encoder == the ORACLE's orc_pq_encode (expansion distances, find_k_neighbors' k = 1
    selection through the pinned heap restatement, packed) — not a numpy copy of the kernel's loop — on tie-heavy
    grid-valued vectors, vectors exactly between two centroids, and vectors whose expansion is all cancellation; sq_dim 3
    (BASELINE configs[4]: no instance in the reference, sequential norms) included; sum_mode 0 likewise."""
    rng = np.random.default_rng(M * 1000 + dim + len(kind))
    n, ds = 8000, dim // M
    if kind == "grid":
        cb, v = rng.integers(0, 3, (M, 16, ds)).astype(np.float32), rng.integers(0, 3, (n, dim)).astype(np.float32)
        cb[:, 7] = cb[:, 2]
    elif kind == "offset":
        cb = (100 + 0.01 * rng.normal(size=(M, 16, ds))).astype(np.float32)
        v = (100 + 0.01 * rng.normal(size=(n, dim))).astype(np.float32)
    else:
        cb, v = rng.normal(size=(M, 16, ds)).astype(np.float32), rng.normal(size=(n, dim)).astype(np.float32)
        cb[3, 7] = cb[3, 2]                                   # duplicate centroid: ties must resolve to the first
    if kind == "mid":
        m = rng.integers(0, M, n)
        ab = np.stack([rng.choice(16, 2, replace=False) for _ in range(n)])
        for i in range(n):
            v[i, m[i] * ds:(m[i] + 1) * ds] = (cb[m[i], ab[i, 0]] + cb[m[i], ab[i, 1]]) * np.float32(0.5)
    want = po.pq_encode(cb, v)
    assert np.array_equal(pyqadc.pq_encode(cb, v), want)
    assert np.array_equal(pyqadc.pq_encode(cb, v, sum_mode=0), po.pq_encode(cb, v, sum_mode=0))
    assert np.array_equal(pyqadc.pq_encode(cb, v, encode_form=0), po.pq_encode(cb, v, form=0))


@pytest.mark.parametrize("M", [16, 32])
def test_ivf_range_shards_merge_by_assign_slot(pyqadc, po, M):
    """Multi-GPU IVF as 3 virtual ranks on one GPU: every partition (labelled, ragged, some so small that a rank
    holds none of it) is range-sharded; each rank carries the partitions' starts; the streams, interleaved by
    (assign slot, rank), replay to the heaps of the unsharded database."""
    from pyqadc import sharded
    rng = np.random.default_rng(7 + M)
    sizes = [20011, 37, 6000, 16, 9003]
    keep, R, nq, ma, world = 0.05, 100, 4, 3, 3
    parts = [rand_codes(rng, s, M) for s in sizes]
    perm = rng.permutation(sum(sizes)).astype(np.uint32)
    labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
    ranks = []
    for r in range(world):
        idx = pyqadc.Index(M)
        for p, s in enumerate(sizes):
            first, ln = sharded.shard_ranges(s, world)[r]
            st = po.start_size(s, keep)
            idx.add_partition_shard(parts[p][first:first + ln], first, s, labels=labels[p][first:first + ln] if ln else None,
                                    starts=parts[p][:st])
        idx.finalize(keep)
        ranks.append(idx)
    assign = np.stack([rng.choice(len(sizes), ma, replace=False) for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M)
    outs = [idx.query_scan_shard_streams(assign, tables.copy(), R) for idx in ranks]
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        ks, vs = [], []
        for slot in range(ma):
            for o in outs:
                assert o["qmax"][q] == np.float32(want["qmax"])
                a, b = int(o["offsets"][q]), int(o["offsets"][q + 1])
                sel = o["slots"][a:b] == slot
                ks.append(o["keys"][a:b][sel])
                vs.append(o["vals"][a:b][sel])
        got = pyqadc.replay_i8(np.concatenate(ks), np.concatenate(vs), R, sentinel=True)
        assert heaps_equal(got, (want["keys"], want["values"])), q
    for idx in ranks:
        idx.close()


def test_borrowed_device_partitions_key_base_and_scan_start(pyqadc, po):
    """Partitions whose codes/labels already live in device memory (torch tensors here), a key offset for an
    unlabelled partition, and the stand-alone pre-scan entry point."""
    import torch
    rng = np.random.default_rng(51)
    M, R, keep = 16, 100, 0.03
    sizes = [30000, 12001]
    parts = [rand_codes(rng, s, M) for s in sizes]
    labels = [rng.permutation(s).astype(np.uint32) + 10 * s for s in sizes]
    tables = float_tables(rng, 2, 2, M)
    assign = np.array([[0, 1], [1, 0]], np.int32)
    # labelled, borrowed
    idx = pyqadc.Index(M)
    keep_alive = []
    for c, l in zip(parts, labels):
        dc = torch.zeros(c.size + 64, dtype=torch.uint8, device="cuda")
        dc[:c.size] = torch.from_numpy(c.reshape(-1)).cuda()
        dl = torch.from_numpy(l.astype(np.int64)).cuda().to(torch.int32)        # same 32 bits as the uint32 labels
        torch.cuda.synchronize()
        keep_alive += [dc, dl]
        idx.add_partition_device(dc.data_ptr(), c.shape[0], dl.data_ptr(), keepalive=(dc, dl))
    idx.finalize(keep)
    res = idx.query_scan(assign, tables.copy(), R)
    qmax = idx.scan_start(assign, tables, R)
    for q in range(2):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"]))
        assert qmax[q] == np.float32(want["qmax"])
    idx.close()
    # unlabelled with a key offset: keys = key_base + position
    idx = pyqadc.Index(M)
    idx.add_partitions([parts[0]])
    idx.set_key_base(0, 1000000)
    idx.finalize(keep)
    t1 = float_tables(rng, 1, 1, M)
    got = idx.query_scan(np.zeros((1, 1), np.int32), t1.copy(), R)["heaps"][0]
    want = po.query_scan(M, [parts[0]], None, keep, [0], t1[0].copy(), R)
    wk = want["keys"].copy()
    real = np.ones(len(wk), bool)
    if want["values"][0] == 127 or (len(wk) < R):              # a surviving (0,127) sentinel keeps key 0
        real = ~((want["keys"] == 0) & (want["values"] == 127))
    wk[real] += 1000000
    assert np.array_equal(got[0], wk) and np.array_equal(got[1], want["values"])
    idx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_quantizer_modes_match_oracle(pyqadc, po, mode):
    rng = np.random.default_rng(60 + mode)
    codes = rand_codes(rng, 50000, 16)
    idx = pyqadc.Index(16)
    idx.add_partitions([codes])
    idx.finalize(0.02)
    idx.set_option("quant_mode", mode)
    tables = float_tables(rng, 4, 1, 16, scale=0.37)
    res = idx.query_scan(np.zeros((4, 1), np.int32), tables.copy(), 100, want_qtables=True)
    for q in range(4):
        want = po.query_scan(16, [codes], None, 0.02, [0], tables[q].copy(), 100, quant_mode=mode)
        assert np.array_equal(res["qtables"][q], want["qtables"])
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"]))
    idx.close()


def test_candidate_output_capacity_error(pyqadc):
    rng = np.random.default_rng(61)
    idx = pyqadc.Index(16)
    idx.add_partitions([rand_codes(rng, 20000, 16)])
    idx.finalize(0.05)
    with pytest.raises(pyqadc.QadcError, match="too small"):
        idx.query_scan_candidates(np.zeros((1, 1), np.int32), float_tables(rng, 1, 1, 16), 100, capacity=4)
    idx.close()


@pytest.mark.parametrize("M,wgs,share,mq", [(16, 0, 0x41, 0), (16, 12, 0x41, 0), (16, 16, 0x49, 0), (32, 0, 0x41, 0),
                                            (16, 0, 0, 0), (16, 0, 0x41, 1), (32, 0, 0x41, 1), (16, 12, 0x41, 1)])
def test_shared_launch_of_a_batch_matches_oracle(pyqadc, po, M, wgs, share, mq):
    """The queries of a batch over the same codes run as L2-sharing siblings (1-D sibling-major launch, XCD decode
    when the workgroups per run are a multiple of 8, plain decode otherwise): same heaps as one query at a time."""
    rng = np.random.default_rng(70 + M + wgs)
    n, nq, R, keep = 700001, 5, 100, 0.01
    codes = rand_codes(rng, n, M)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(keep)
    idx.set_option("small_run", 32768)          # streaming kernel from the 32768-code level on
    idx.set_option("wgs_per_item", wgs)
    idx.set_option("share_variant", share)
    idx.set_option("mq", mq)
    tables = float_tables(rng, nq, 1, M)
    res = idx.query_scan(np.zeros((nq, 1), np.int32), tables.copy(), R)
    for q in range(nq):
        want = po.query_scan(M, [codes], None, keep, [0], tables[q].copy(), R)
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


@pytest.mark.parametrize("M", [16, 32])
def test_multi_query_pass_groups_labels_and_padding_replays(pyqadc, po, M):
    """19 queries = groups of 8, 8 and 3 sharing one pass each; labelled ragged partition (padding-lane replays of the
    last code), tie-heavy tables, tiny candidate regions (regrow path) — heaps equal the one-query-at-a-time oracle."""
    rng = np.random.default_rng(80 + M)
    n, nq, R, keep = 300003, 19, 37, 0.02
    codes = rand_codes(rng, n, M)
    labels = (rng.permutation(n).astype(np.uint32) * 7 + 3).astype(np.uint32)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes], labels=[labels])
    idx.finalize(keep)
    idx.set_option("small_run", 8192)
    idx.set_option("cand_capacity", 256)
    tables = float_tables(rng, nq, 1, M, scale=0.2)
    tables = np.round(tables * 3) / 3                          # few distinct values: massive ties
    res = idx.query_scan(np.zeros((nq, 1), np.int32), tables.copy(), R)
    for q in range(nq):
        want = po.query_scan(M, [codes], [labels], keep, [0], tables[q].copy(), R)
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
    # same through the direct int8 entry point, one query per table, R > candidates of the early levels
    qt = rng.integers(0, 9, (nq, 1, M, 16)).astype(np.int8)
    heaps = idx.scan_i8(np.zeros((nq, 1), np.int32), qt, R)
    for q in range(nq):
        want = po.scan_i8(M, [codes], [labels], qt[q, 0], R)
        assert heaps_equal(heaps[q], want), q
    idx.close()


@pytest.mark.parametrize("M,W,sample", [(16, 3, 65536), (16, 8, 512), (32, 2, 512), (16, 5, 64)])
def test_sharded_prescan_equals_unsharded(pyqadc, po, M, W, sample):
    """Multi-GPU pre-scan protocol on one GPU: W slices of the starts are pre-scanned separately, the R smallest values
    of each are concatenated (what the all-gather does) and injected; qmax, tables and heaps equal the plain path and
    the oracle.  IVF-shaped (3 probes, ragged partitions, one with fewer starts than slices)."""
    rng = np.random.default_rng(90 + M + W)
    R, keep, nq, ma = 50, 0.04, 6, 3
    sizes = [90000, 1234, 40000, 17, 25000]
    parts = [rand_codes(rng, n, M) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(keep)
    idx.set_option("prescan_sample", sample)
    idx.set_option("small_run", 4096)
    assign = np.stack([rng.permutation(len(sizes))[:ma] for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M)
    plain = idx.query_scan(assign, tables.copy(), R, want_qtables=True)
    gathered = []
    for r in range(W):
        idx.prescan_submit(r % 2, assign, tables.copy(), R, r, W)
        gathered.append(idx.prescan_collect(r % 2))
    gathered = np.concatenate(gathered, axis=1)                       # [nq][W*R]
    tb = tables.copy()
    idx.submit(0, assign, tb, R, prescan=gathered)
    got = idx.collect(0)
    for q in range(nq):
        want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
        assert heaps_equal((got["keys"][q, :got["sizes"][q]], got["values"][q, :got["sizes"][q]]),
                           (want["keys"], want["values"])), q
        assert heaps_equal((got["keys"][q, :got["sizes"][q]], got["values"][q, :got["sizes"][q]]), plain["heaps"][q]), q
    # fewer than R starts in total: the sentinel survives and every rank reports it
    idx2 = pyqadc.Index(M)
    idx2.add_partitions([parts[1]])
    idx2.finalize(0.01)                                               # 12 starts < R
    a1 = np.zeros((2, 1), np.int32)
    t1 = float_tables(rng, 2, 1, M)
    pv = []
    for r in range(W):
        idx2.prescan_submit(0, a1, t1.copy(), R, r, W)
        pv.append(idx2.prescan_collect(0))
    pv = np.concatenate(pv, axis=1)
    assert np.sum(pv < 3e38) == 2 * 12
    idx2.submit(1, a1, t1.copy(), R, prescan=pv)
    assert list(idx2.collect(1)["status"]) == [1, 1]
    idx.close()
    idx2.close()


@pytest.mark.parametrize("M,W", [(16, 4), (32, 3)])
def test_multi_gpu_protocol_on_virtual_ranks(pyqadc, po, M, W):
    """The whole multi-GPU step on one GPU: W shard indexes (contiguous code ranges + a replica of the starts),
    sliced pre-scan -> gather -> injected submit -> candidate streams -> native merge in (rank, position) order.
    Heaps equal the oracle's single sequential scan of the whole list."""
    from pyqadc import sharded
    rng = np.random.default_rng(95 + M + W)
    n, nq, R, keep = 260003, 9, 100, 0.03
    codes = rand_codes(rng, n, M)
    starts = max(1, int(np.float32(n) * np.float32(keep)))
    ranges = sharded.shard_ranges(n, W)
    ranks = []
    for first, ln in ranges:
        ix = pyqadc.Index(M)
        ix.add_partition_shard(codes[first:first + ln], first, n, starts=codes[:starts])
        ix.finalize(keep)
        ix.set_option("small_run", 4096)
        ranks.append(ix)
    assign = np.zeros((nq, 1), np.int32)
    tables = float_tables(rng, nq, 1, M)
    pv = []
    for r, ix in enumerate(ranks):
        ix.prescan_submit(0, assign, tables.copy(), R, r, W)
        pv.append(ix.prescan_collect(0))
    gathered = np.concatenate(pv, axis=1)
    cap = 1 << 14
    bufs = []
    for ix in ranks:
        ix.submit(0, assign, tables.copy(), R, prescan=gathered)
        res = ix.collect_candidates(0)
        words = sharded._layout(nq, cap, 0, 1)[2]
        bufs.append(sharded.pack_stream(np.zeros(words, np.int32), res, nq, cap))
    K, V, S = np.zeros((nq, R), np.uint32), np.zeros((nq, R), np.int8), np.zeros(nq, np.int32)
    pyqadc.merge_streams_i8(np.stack(bufs), W, nq, R, cap, 1, 0, 1, None, K, V, S)
    for q in range(nq):
        want = po.query_scan(M, [codes], None, keep, [0], tables[q].copy(), R)
        assert heaps_equal((K[q, :S[q]], V[q, :S[q]]), (want["keys"], want["values"])), q
    for ix in ranks:
        ix.close()


def test_pipelined_slots_with_regrows_and_changing_batch_shapes(pyqadc, po):
    """Three batches in flight, every batch a different shape (queries, R), candidate regions far too small (every
    batch is regrown and re-run inside collect while the others are still on the GPU): all heaps equal the oracle."""
    rng = np.random.default_rng(123)
    M, keep = 16, 0.02
    sizes = [120000, 3000, 45001]
    parts = [rand_codes(rng, n, M) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(keep)
    idx.set_option("wgq", 0)                # this test is about the level-structured path's own fallbacks
    idx.set_option("head_level", 0)         # (with every bound level launched separately)
    idx.set_option("small_run", 8192)
    idx.set_option("cand_capacity", 32)
    batches = []
    for b in range(9):
        nq = int(rng.integers(1, 13))
        ma = int(rng.integers(1, 4))
        R = int(rng.choice([5, 50, 100]))
        if b % 2 == 0:
            assign = np.tile(rng.permutation(3)[:ma], (nq, 1)).astype(np.int32)       # shared probes: multi-query launches
        else:
            assign = np.stack([rng.permutation(3)[:ma] for _ in range(nq)]).astype(np.int32)
        batches.append((assign, float_tables(rng, nq, ma, M), R))
    pending, results = [], {}
    for b, (assign, tables, R) in enumerate(batches):
        idx.submit(b % 3, assign, tables.copy(), R)
        pending.append(b)
        if len(pending) == 3:
            d = pending.pop(0)
            results[d] = idx.collect(d % 3)
    while pending:
        d = pending.pop(0)
        results[d] = idx.collect(d % 3)
    assert idx.profile()["regrows"] >= 3                       # each slot outgrew its 32-entry regions at least once
    for b, (assign, tables, R) in enumerate(batches):
        r = results[b]
        for q in range(assign.shape[0]):
            want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
            n = int(r["sizes"][q])
            assert heaps_equal((r["keys"][q, :n], r["values"][q, :n]), (want["keys"], want["values"])), (b, q)
    idx.close()


# ---------------------------------------------------------------------------------------------------------------
# one workgroup per query (qadc_query_kernel.hip): the path IVF batches and small lists take
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32])
def test_wgq_stream_capacity_regrow_is_exact(pyqadc, po, M):
    rng = np.random.default_rng(80 + M)
    parts = [rand_codes(rng, n, M) for n in (90001, 37, 20000)]
    labels = [rng.permutation(1 << 20)[:len(p)].astype(np.uint32) for p in parts]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.01)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_capacity", 16)       # far too small: overflow -> every entry counted -> regrown -> re-run
    idx.set_option("profile", 1)
    assign = np.array([[0, 1, 2], [2, 0, 1], [1, 1, 0]], np.int32)
    qt = rand_qtables(rng, (3, 3), M, 20 if M == 16 else 7)
    got = idx.scan_i8(assign, qt, 100)
    pr = idx.profile()
    assert pr["regrows"] >= 1 and pr["wgq_launches"] >= 1 and pr["scan_launches"] == 0
    for q in range(3):
        want = po.scan_i8(M, [parts[p] for p in assign[q]], [labels[p] for p in assign[q]], qt[q], 100)
        assert heaps_equal(got[q], want), q
    idx.close()


@pytest.mark.gpu
def test_grouped_second_phase_orders_more_than_4096_candidates_per_query(pyqadc, po):
    """Partition-major second phase with a deliberately loose bound (head of ONE small partition, then five probes of
    9000 codes whose share below that bound runs into the thousands): some queries end up with 4097 .. 8192 candidates,
    which the ordering pass of this path takes (order_cands_kernel: 8192 sort keys in LDS, 8 entries per thread) where the
    query kernel's own tail stops at 4096 — no fallback, heaps equal the oracle's."""
    rng = np.random.default_rng(4242)
    M, nq, ma, R, keep = 16, 72, 6, 100, 0.05
    sizes = [400] + [9000] * 11
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.permutation(1 << 22)[:n].astype(np.uint32) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    for k, v in (("wgq", 2), ("wgq_group", 2), ("wgq_group_head", 1), ("device_replay_alone_nq", 0), ("profile", 1)):
        idx.set_option(k, v)                               # (device_replay_alone_nq: the grouped phase needs the device replay)
    assign = np.stack([np.concatenate([[0], rng.permutation(np.arange(1, 12))[:ma - 1]]) for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M, scale=1.0)
    tables[:, 1:, :] *= np.float32(1.3)                    # (later probes mostly above the head's values: tuned so that no query passes 8192)
    res = idx.query_scan(assign, tables.copy(), R)
    pr = idx.profile()
    assert pr["group_launches"] >= 1 and pr["group_fallbacks"] == 0, pr
    estimate = []                                          # candidates per query under a bound drawn from the head alone
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        assert want["rc"] == res["status"][q]
        if want["rc"]:
            continue
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
        qt = want["qtables"].reshape(ma, M * 16)
        head = po.candidates_i8(M, parts[0], qt[0]).astype(np.int32)
        hs = np.sort(head[head < 127])
        bound = hs[R - 2] if len(hs) >= R - 1 else 127
        estimate.append(int((head < 127).sum()) + sum(int((po.candidates_i8(M, parts[assign[q][a]], qt[a]).astype(np.int32) < bound).sum())
                                                        for a in range(1, ma)))
    assert max(estimate) > 5000 and sum(e > 4096 for e in estimate) >= 2, sorted(estimate)[-5:]   # (what the case is for)
    # ... and with the ordering pass held to the query kernel's 4096 the same batch does fall back (and still ends right)
    idx.set_option("wgq_cand_cap", 4096)
    idx.profile_reset()
    res2 = idx.query_scan(assign, tables.copy(), R)
    assert idx.profile()["group_fallbacks"] >= 1
    for q in range(nq):
        assert res2["status"][q] == res["status"][q]
        if res["status"][q] == 0:
            assert heaps_equal(res2["heaps"][q], res["heaps"][q]), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [100, 288, 289, 320, 321])
def test_device_replay_of_a_query_kernel_batch_at_the_heap_sizes_around_its_limits(pyqadc, po, R):
    """70 queries through the one-workgroup-per-query path with the heap replay on the device (>= 64 queries): R up to
    the lane replay's 288 / the wave replay's 320 elements stays on the GPU, R = 321 is replayed on the host — heaps
    equal the oracle's either way (the scan_path fixture runs this with both device replays)."""
    rng = np.random.default_rng(900 + R)
    M, nq = 16, 70
    parts = [rand_codes(rng, n, M) for n in (30011, 4001, 9000)]
    labels = [rng.permutation(1 << 20)[:len(p)].astype(np.uint32) for p in parts]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.05)
    idx.set_option("wgq", 2)
    assign = np.stack([rng.permutation(3)[:2] for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, 2, M, scale=0.3)
    res = idx.query_scan(assign, tables.copy(), R)
    for q in range(nq):
        want = po.query_scan(M, parts, labels, 0.05, assign[q], tables[q].copy(), R)
        assert want["rc"] == res["status"][q]
        if want["rc"] == 0:
            assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), (R, q)
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,n,keep", [(16, 2000003, 0.01), (32, 1200001, 0.02)])
def test_wgq_prescan_values_beyond_the_lds_budget(pyqadc, po, M, n, keep):
    """More starts than the kernel keeps in LDS (12288 / 8192 values): the values go through the global scratch and
    the select reads them from there — qmin, qmax, int8 tables and heaps must still equal the oracle's."""
    rng = np.random.default_rng(5 + M)
    codes = rand_codes(rng, n, M)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    tables = float_tables(rng, 2, 1, M)
    got = idx.query_scan(np.zeros((2, 1), np.int32), tables.copy(), 100, want_qtables=True)
    for q in range(2):
        want = po.query_scan(M, [codes], None, keep, [0], tables[q].copy(), 100)
        assert got["qmax"][q] == np.float32(want["qmax"]) and got["qmin"][q] == np.float32(want["qmin"])
        assert np.array_equal(got["qtables"][q], want["qtables"])
        assert heaps_equal(got["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,sizes,keep,R", [(16, (60001, 90001, 8000, 70001), 0.02, 100), (32, (1200001,), 0.02, 100),
                                           (16, (60001, 90001, 70001), 0.02, 1000), (16, (150001,), 0.02, 1)])
def test_front_select_threshold_gives_the_oracles_qmax(pyqadc, po, M, sizes, keep, R):
    """The front's select (R-th smallest pre-scan value = qmax) first keeps the keys below a threshold drawn from 64
    sampled values and runs its radix passes over those.  3 K .. 24 K values (LDS and global scratch), R = 1, 100 and 1000:
    qmin, qmax, int8 tables and heaps equal the oracle's every time."""
    rng = np.random.default_rng(77 + M + len(sizes) + R)
    parts = [rand_codes(rng, n, M) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    nq, ma = 3, len(sizes)
    assign = np.stack([rng.permutation(len(sizes))[:ma] for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M)
    got = idx.query_scan(assign, tables.copy(), R, want_qtables=True)
    for q in range(nq):
        want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
        assert want["rc"] == got["status"][q]
        assert got["qmax"][q] == np.float32(want["qmax"]) and got["qmin"][q] == np.float32(want["qmin"]), (q, got["qmax"][q], want["qmax"])
        assert np.array_equal(got["qtables"][q], want["qtables"])
        if want["rc"] == 0:
            assert heaps_equal(got["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32])
@pytest.mark.parametrize("sample", ["low", "high"])
def test_front_select_with_an_unlucky_sample_falls_back_exactly(pyqadc, po, M, sample):
    """The select's two fallbacks, reached by DATA: the 64 sampled pre-scan values (evenly spaced over the 8000 starts of a flat
    list: positions lane * 8000 / 64) are made all-smallest ("low": the threshold keeps fewer than R keys) or all-largest ("high":
    it keeps nearly everything — more than the kept-key area holds); either way the passes then run over all values and qmax,
    the int8 tables and the heaps are the oracle's."""
    rng = np.random.default_rng(4100 + M + len(sample))
    n, keep, R = 400000, 0.02, 100
    codes = rand_codes(rng, n, M)
    nstart = po.start_size(n, keep)
    assert nstart == 8000
    pos = (np.arange(64, dtype=np.int64) * nstart) // 64
    codes[:nstart] |= 0x11                                      # no start code has a zero nibble ...
    codes[pos] = 0x00 if sample == "low" else 0xff              # ... or an all-ones one, except the sampled ones
    codes[:nstart][np.all(codes[:nstart] == 0xff, axis=1) & ~np.isin(np.arange(nstart), pos)] ^= 0x22
    tables = float_tables(rng, 1, 1, M)
    t = tables.reshape(M, 16)
    t[:, 0] = 0.0                                               # nibble 0: distance 0 — the sampled codes of "low" sum to exactly 0
    t[:, 15] = t.max() * 4 + 1                                  # nibble 15: far — the sampled codes of "high" are the largest sums
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    got = idx.query_scan(np.zeros((1, 1), np.int32), tables.copy(), R, want_qtables=True)
    want = po.query_scan(M, [codes], None, keep, [0], tables[0].copy(), R)
    assert want["rc"] == got["status"][0] == 0
    assert got["qmax"][0] == np.float32(want["qmax"]) and np.array_equal(got["qtables"][0], want["qtables"])
    assert heaps_equal(got["heaps"][0], (want["keys"], want["values"]))
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 7, 100, 320, 321, 1000])
def test_wgq_device_replay_and_host_replay_agree(pyqadc, po, R):
    """Batches of >= 64 queries replay on the device, one wave per query with the heap in registers (replay_heap_wave_kernel,
    R <= 320); larger heaps and small batches replay on the host.  Ragged labelled partitions, tie-heavy tables, ma = 5."""
    rng = np.random.default_rng(R)
    M, K, ma, nq = 16, 12, 5, 130
    sizes = [int(x) for x in rng.integers(1, 9000, K)]
    sizes[3] = 0
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.05)
    idx.set_option("wgq", 2)
    idx.set_option("device_replay_alone_nq", 0)                 # (a synchronous call would otherwise replay on the host below 512 queries)
    assign = np.stack([rng.permutation(K)[:ma] for _ in range(nq)]).astype(np.int32)
    qt = rand_qtables(rng, (nq, ma), M, 9)
    got = idx.scan_i8(assign, qt, R)
    idx.set_option("device_replay_nq", 0)
    got_host = idx.scan_i8(assign, qt, R)
    for q in range(nq):
        want = po.scan_i8(M, [parts[p] for p in assign[q]], [labels[p] for p in assign[q]], qt[q], R)
        assert heaps_equal(got[q], want), q
        assert heaps_equal(got_host[q], want), q
    idx.close()


@pytest.mark.gpu
def test_wgq_pipelined_slots_changing_shapes_with_regrows(pyqadc, po):
    rng = np.random.default_rng(321)
    M, keep = 16, 0.02
    sizes = [120000, 3000, 45001, 16, 0, 70000]
    parts = [rand_codes(rng, n, M) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_capacity", 64)
    idx.set_option("profile", 1)
    batches = []
    for b in range(9):
        nq = int(rng.choice([1, 3, 12, 70]))
        ma = int(rng.integers(1, 5))
        R = int(rng.choice([5, 50, 100]))
        assign = np.stack([rng.permutation(len(sizes))[:ma] for _ in range(nq)]).astype(np.int32)
        batches.append((assign, float_tables(rng, nq, ma, M, negatives=(b % 3 == 0)), R))
    pending, results = [], {}
    for b, (assign, tables, R) in enumerate(batches):
        idx.submit(b % 3, assign, tables, R)
        pending.append(b)
        if len(pending) == 3:
            d = pending.pop(0)
            results[d] = idx.collect(d % 3)
    while pending:
        d = pending.pop(0)
        results[d] = idx.collect(d % 3)
    assert idx.profile()["regrows"] >= 3
    for b, (assign, tables, R) in enumerate(batches):
        res = results[b]
        for q in range(assign.shape[0]):
            # the submitted tables were clamped in place by collect; the oracle clamps its own copy the same way
            want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
            if want["rc"] != 0:
                assert res["status"][q] == 1
                continue
            sz = res["sizes"][q]
            assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), (want["keys"], want["values"])), (b, q)
    idx.close()


@pytest.mark.gpu
def test_wgq_search_keeps_assign_on_the_device(pyqadc, po):
    """qadc_search on the one-workgroup-per-query path: coarse assignment, tables, pre-scan, quantizer, scan and (for
    >= 64 queries) the heap replay all stay on the GPU; assign[] only comes back for the caller.  Equal to the
    level-structured path of the same index, query for query."""
    rng = np.random.default_rng(77)
    M, K, ma, nq, dim = 16, 40, 6, 96, 64
    sizes = [int(x) for x in rng.integers(200, 6000, K)]
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.permutation(1 << 22)[:n].astype(np.uint32) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(0.03)
    idx.set_pq(rng.normal(size=(M, 16, dim // M)).astype(np.float32))
    idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    idx.set_option("wgq", 2)
    a = idx.search(queries, ma, 100)
    idx.set_option("wgq", 0)
    b = idx.search(queries, ma, 100)
    assert np.array_equal(a["assign"], b["assign"]) and np.array_equal(a["status"], b["status"])
    for q in range(nq):
        assert heaps_equal(a["heaps"][q], b["heaps"][q]), q
    idx.close()


@pytest.mark.gpu
def test_wgq_candidate_overflow_falls_back_to_the_level_path(pyqadc, po):
    """A query with more candidates than a workgroup sorts in LDS (here: cap forced to 64; in production 4096, e.g. an
    all-zero table where every code ties) is re-run through the level-structured path; heaps stay exact."""
    rng = np.random.default_rng(99)
    M = 16
    parts = [rand_codes(rng, n, M) for n in (50001, 7000)]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts)
    idx.finalize(0.01)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_cand_cap", 64)
    idx.set_option("profile", 1)
    assign = np.array([[0, 1], [1, 0]], np.int32)
    qt = rand_qtables(rng, (2, 2), M, 15)
    qt[1] = 0                                              # every code of query 1 sums to 0
    got = idx.scan_i8(assign, qt, 100)
    pr = idx.profile()
    assert pr["regrows"] >= 1 and pr["scan_launches"] + pr["small_launches"] >= 1
    for q in range(2):
        assert heaps_equal(got[q], po.scan_i8(M, [parts[p] for p in assign[q]], None, qt[q], 100)), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32])
def test_blas_expansion_tables_go_negative_and_are_clamped_like_the_reference(pyqadc, po, M):
    """Real-encoded data, queries that nearly coincide with database vectors: the BLAS-expansion tables
    (distances.hpp:151-183; what nns_engine_batch / ma > 1 evaluate) contain slightly NEGATIVE entries, so
    query_scan's `qmin < 0 -> qmin = 0, clamp in place` branch (db_query_4.cpp:258-269) runs inside the engine —
    on the device for qadc_search, on the host for the C-ABI's float-table entry point — with the oracle's result."""
    rng = np.random.default_rng(500 + M)
    dim, n, nq, R, keep = 64, 40000, 24, 50, 0.02
    ds = dim // M
    base = (rng.normal(size=(n, dim)) * 3).astype(np.float32)
    cb = np.stack([base[rng.integers(0, n, 16), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
    codes = pyqadc.pq_encode(cb, base)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(keep)
    idx.set_pq(cb)
    idx.set_option("table_form", 1)
    # queries = concatenations of codebook entries (what a decoded database vector is) + a hair of noise
    pick = rng.integers(0, 16, (nq, M))
    queries = np.concatenate([cb[m][pick[:, m]] for m in range(M)], axis=1).astype(np.float32)
    queries *= (1 + rng.normal(size=queries.shape).astype(np.float32) * np.float32(3e-4))
    res = idx.search(queries, 1, R)
    negatives = 0
    for q in range(nq):
        tables = _seq_expansion(queries[q].reshape(M, 1, ds), cb).reshape(1, M * 16)
        negatives += int((tables < 0).any())
        mine = np.ascontiguousarray(tables.copy())
        want = po.query_scan(M, [codes], None, keep, [0], mine, R)
        assert want["rc"] == res["status"][q] == 0
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q
        if (tables < 0).any():
            assert want["qmin"] == 0.0 and (mine >= 0).all()        # the oracle clamped its copy in place
            # the float-table entry point clamps the caller's buffer the same way and ends in the same heap
            t2 = np.ascontiguousarray(tables.copy())
            r2 = idx.query_scan(np.zeros((1, 1), np.int32), t2.reshape(1, 1, M * 16), R)
            assert r2["qmin"][0] == 0.0 and np.array_equal(t2.reshape(-1), mine.reshape(-1))
            assert heaps_equal(r2["heaps"][0], (want["keys"], want["values"]))
    assert negatives >= nq // 4, negatives
    idx.close()


# ---------------------------------------------------------------------------------------------------------------
# native multi-GPU merge (qadc_dist_*): one ncclAllGather + device-side replay in global scan order
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,world", [(16, 3), (32, 4), (16, 8)])
def test_dist_merge_kernel_on_virtual_ranks(pyqadc, po, M, world):
    """The device half of qadc_dist_collect (dist_merge_lanes_kernel) on the blocks `world` virtual ranks of ONE GPU
    would contribute to the all-gather: every partition (labelled, ragged, some so small that a rank holds none of it)
    range-sharded, tie-heavy tables, ma = 3: the merged heaps equal the unsharded oracle's, array for array."""
    from pyqadc import sharded
    rng = np.random.default_rng(70 + M + world)
    sizes = [20011, 37, 6000, 16, 9003, 0, 12345]
    keep, R, nq, ma = 0.05, 100, 70, 3
    parts = [rand_codes(rng, s, M) for s in sizes]
    perm = rng.permutation(sum(sizes)).astype(np.uint32)
    labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
    assign = np.stack([rng.choice([0, 1, 2, 3, 4, 6], ma, replace=False) for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M, scale=0.3)
    streams = []
    for r in range(world):
        idx = pyqadc.Index(M)
        for p, s_ in enumerate(sizes):
            if s_ == 0:
                idx.add_partitions([np.zeros((0, M // 2), np.uint8)], [np.zeros(0, np.uint32)])
                continue
            first, ln = sharded.shard_ranges(s_, world)[r]
            st = po.start_size(s_, keep)
            idx.add_partition_shard(parts[p][first:first + ln], first, s_, labels=labels[p][first:first + ln] if ln else None,
                                    starts=parts[p][:st])
        idx.finalize(keep)
        streams.append(idx.query_scan_shard_streams(assign, tables.copy(), R))
        idx.close()
    got = pyqadc.dist_merge_blocks(streams, nq, ma, R)
    # ... and the host half: the share replay the ranks of a few-query batch (bench.py's 32-query steps) run between
    # their two all-gathers, all `world` shares in turn
    got_host = pyqadc.dist_merge_blocks(streams, nq, ma, R, host=True)
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        assert want["rc"] == 0
        assert heaps_equal(got[q], (want["keys"], want["values"])), q
        assert heaps_equal(got_host[q], (want["keys"], want["values"])), q


@pytest.mark.gpu
@pytest.mark.parametrize("world,ma,seed", [(8, 64, 1), (5, 37, 2), (16, 9, 3), (2, 200, 4)])
def test_dist_interleave_run_bounds_on_synthetic_streams(pyqadc, world, ma, seed):
    """dist_interleave_kernel derives every (rank, slot) count from the BOUNDS of the slot's run in the rank's stream (a
    stream is sorted by slot) — the lanes at a run's first and last entry, neighbours compared by shuffle, rows of 64 entries,
    four rows in flight.  Synthetic streams that lean on exactly that: runs of one entry, runs that start or end on a row
    boundary (i = 63 / 64 / 255 / 256 / 1023 / 1024), slots missing on some ranks, empty streams, streams of a single
    run, up to 3000 entries per rank and query.  The device merge must give what the host-share replay (an independent
    implementation: qadc_dist_merge_blocks_host) gives from the same blocks, heap array for heap array."""
    rng = np.random.default_rng(9000 + seed)
    nq, R = 19, 100
    streams = []
    for g in range(world):
        keys, vals, slots, offsets = [], [], [], [0]
        for q in range(nq):
            kind = (q + g) % 6
            if kind == 0:
                lens = np.zeros(ma, np.int64)                                   # an empty stream
            elif kind == 1:
                lens = np.zeros(ma, np.int64)
                lens[rng.integers(ma)] = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025]))   # one run
            elif kind == 2:
                lens = np.ones(ma, np.int64)                                    # every run a single entry
                lens[rng.random(ma) < 0.3] = 0
            else:
                lens = rng.integers(0, 2 * (3000 // ma) + 2, ma)
                lens[rng.random(ma) < 0.25] = 0
                # pad the first runs so that later ones start exactly on row boundaries
                for target in (64, 256, 1024):
                    c = np.cumsum(lens)
                    j = int(np.searchsorted(c, target))
                    if j < ma and c[j] >= target and lens[j] > 0:
                        lens[j] -= c[j] - target
            n = int(lens.sum())
            slots.append(np.repeat(np.arange(ma), lens))
            keys.append(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32))
            vals.append(rng.integers(0, 127, n).astype(np.int8))
            offsets.append(offsets[-1] + n)
        streams.append(dict(keys=np.concatenate(keys), vals=np.concatenate(vals), slots=np.concatenate(slots).astype(np.uint32),
                            offsets=np.asarray(offsets, np.uint64)))
    got = pyqadc.dist_merge_blocks(streams, nq, ma, R)
    want = pyqadc.dist_merge_blocks(streams, nq, ma, R, host=True)
    for q in range(nq):
        assert np.array_equal(got[q][0], want[q][0]) and np.array_equal(got[q][1], want[q][1]), q
    # the run-bound counts assume a stream grouped by ASCENDING slot below ma: one that is not is refused, not scattered out of
    # place — a slot run split in two, and a slot beyond ma
    if ma > 1 and seed == 0:
        for how in ("split_run", "slot_out_of_range"):
            bad = [dict(st, slots=st["slots"].copy()) for st in streams]
            g = next(g for g in range(world) if len(streams[g]["slots"]) > 8)
            off = streams[g]["offsets"]
            q = int(np.argmax(np.diff(off.astype(np.int64))))            # the rank's longest stream
            a, b = int(off[q]), int(off[q + 1])
            if how == "split_run":
                if bad[g]["slots"][a] == bad[g]["slots"][b - 1]:
                    continue                                              # (a single run: nothing to split)
                bad[g]["slots"][b - 1] = bad[g]["slots"][a]               # the first slot comes back at the end
            else:
                bad[g]["slots"][b - 1] = ma                               # one past the probes
            with pytest.raises(pyqadc.QadcError, match="not grouped by assign slot"):
                pyqadc.dist_merge_blocks(bad, nq, ma, R)


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 2, 3, 31, 62, 63, 64, 65, 100, 126, 127, 128, 129, 190, 191, 192, 254, 255, 256, 300, 318, 319, 320])
def test_wave_replay_kernel_against_the_heap_oracle(pyqadc, po, R):
    """replay_heap_wave_kernel alone (through qadc_dist_merge_blocks with a world of one rank): arbitrary push streams —
    empty, shorter than R, exactly R, long; constant, ascending, descending, two-valued and uniform values (ties at
    every level of the heap) — end in the array kv_binheap::push leaves (binheap.hpp:75-116), for every register count
    of the wave heap and the R values either side of each 64-position boundary."""
    rng = np.random.default_rng(4000 + R)
    lens = [0, 1, max(R - 2, 0), R - 1, R, R + 1, 2 * R + 3, 700, 3000, 6000]
    kinds = ["uniform", "const", "asc", "desc", "two", "narrow", "sawtooth"]
    keys, vals, offs = [], [], [0]
    for n in lens:
        for kind in kinds:
            if kind == "uniform":
                v = rng.integers(0, 128, n)
            elif kind == "const":
                v = np.full(n, 77)
            elif kind == "asc":
                v = np.sort(rng.integers(0, 128, n))
            elif kind == "desc":
                v = np.sort(rng.integers(0, 128, n))[::-1]
            elif kind == "two":
                v = rng.choice([40, 41], n)
            elif kind == "narrow":
                v = rng.integers(60, 64, n)
            else:
                v = 127 - (np.arange(n) % 128)
            keys.append(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))
            vals.append(v.astype(np.int8))
            offs.append(offs[-1] + n)
    nq = len(keys)
    st = dict(keys=np.concatenate(keys), vals=np.concatenate(vals), slots=np.zeros(offs[-1], np.uint32), offsets=np.array(offs, np.int64))
    got = pyqadc.dist_merge_blocks([st], nq, 1, R)
    for q in range(nq):
        want = po.heap_replay_i8(np.concatenate([[0], keys[q]]).astype(np.uint32), np.concatenate([[127], vals[q]]).astype(np.int8), R)
        assert np.array_equal(got[q][0], want[0]) and np.array_equal(got[q][1], want[1]), (R, q, lens[q // len(kinds)], kinds[q % len(kinds)])


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl", "shm"])
@pytest.mark.parametrize("replay", ["host_share", "device_lanes"])
@pytest.mark.parametrize("shape", ["flat_levels", "flat_small", "ivf"])
def test_dist_collect_world_1_over_rccl(pyqadc, po, shape, replay, transport):
    """qadc_dist_init / qadc_dist_collect end to end with a world of ONE rank over real RCCL (all a 1-GPU box allows;
    world > 1: tests/test_gpu_dist_multiproc.py) and over the shared-memory transport:
    pack kernel -> all-gather -> merge kernel -> heaps; the extra payload comes back; a deliberately tiny gather block
    is regrown; results equal the plain collect of the same batches — and a batch submitted AFTER qadc_dist_init may
    still be collected by the plain calls (synchronous query_scan included): its streams are fetched and replayed on
    the host."""
    rng = np.random.default_rng(len(shape))
    M, R, keep = 16, 100, 0.01
    if shape == "flat_levels":
        parts, labels, nq, ma = [rand_codes(rng, 600000, M)], None, 5, 1
    elif shape == "flat_small":
        parts, labels, nq, ma = [rand_codes(rng, 50001, M)], None, 3, 1
    else:
        sizes = [int(x) for x in rng.integers(100, 7000, 30)]
        parts = [rand_codes(rng, n, M) for n in sizes]
        labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
        nq, ma = 80, 4
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    assign = np.stack([rng.permutation(len(parts))[:ma] for _ in range(nq)]).astype(np.int32)
    tables = float_tables(rng, nq, ma, M)
    idx.submit(0, assign, tables.copy(), R)
    plain = idx.collect(0)
    tr = None
    if transport == "rccl":
        idx.dist_init(0, 1, pyqadc.dist_unique_id())
    else:
        import uuid
        tr = pyqadc.ShmTransport("/qadc_w1_%s" % uuid.uuid4().hex[:10], 0, 1, slot_bytes=16 << 20)
        idx.dist_init_transport(tr)
    extra = rng.random(37).astype(np.float32)
    # few-query batches replay their share of the gathered streams on the host and exchange heaps with a second,
    # tiny all-gather; large ones replay everything on the device, one lane per query: both must give the same heaps
    idx.set_option("dist_device_nq", 1 if replay == "device_lanes" else 1 << 20)
    idx.set_option("dist_cap_entries", 64)            # far too small: every rank sees the overflow in the gathered headers
    idx.set_option("profile", 1)                      # and regrows its block alike; the gather is repeated
    for attempt in range(2):
        idx.submit(attempt, assign, tables.copy(), R)
        got = idx.dist_collect(attempt, extra=extra)
        assert np.array_equal(got["extra"], extra[None, :])
        assert np.array_equal(got["sizes"], plain["sizes"]) and np.array_equal(got["status"], plain["status"])
        for q in range(nq):
            sz = plain["sizes"][q]
            assert np.array_equal(got["keys"][q, :sz], plain["keys"][q, :sz]) and np.array_equal(got["values"][q, :sz], plain["values"][q, :sz]), q
    assert idx.profile()["regrows"] >= 1
    # plain collects of batches submitted under the merge: asynchronous pair and the synchronous call
    idx.submit(2, assign, tables.copy(), R)
    again = idx.collect(2)
    sync = idx.query_scan(assign, tables.copy(), R)
    stream = idx.query_scan_shard_streams(assign, tables.copy(), R)
    assert np.array_equal(again["sizes"], plain["sizes"]) and np.array_equal(sync["sizes"], plain["sizes"])
    for q in range(nq):
        sz = plain["sizes"][q]
        assert np.array_equal(again["keys"][q, :sz], plain["keys"][q, :sz]) and np.array_equal(again["values"][q, :sz], plain["values"][q, :sz]), q
        assert np.array_equal(sync["keys"][q, :sz], plain["keys"][q, :sz]) and np.array_equal(sync["values"][q, :sz], plain["values"][q, :sz]), q
        lo, hi = int(stream["offsets"][q]), int(stream["offsets"][q + 1])
        if plain["status"][q] == 0:
            assert heaps_equal(pyqadc.replay_i8(stream["keys"][lo:hi], stream["vals"][lo:hi], R, sentinel=True),
                               (plain["keys"][q, :sz], plain["values"][q, :sz])), q
    idx.close()
    if tr is not None:
        tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,split", [(16, 8), (16, 3), (32, 5)])
def test_wgq_small_batches_split_a_query_over_several_workgroups(pyqadc, po, M, split):
    """A batch too small to fill the GPU gives every query several workgroups: each tightens its bound on the query's
    first block (only workgroup 0 emits from it), scans its own chunk of the rest, and the sub-streams are concatenated
    in workgroup order.  Labelled ragged partitions, ma = 4, chunks that cut partitions in the middle: heaps == oracle."""
    rng = np.random.default_rng(40 + M + split)
    sizes = [70001, 37, 52000, 16, 90003, 0, 33333]
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
    keep, R, nq, ma = 0.01, 100, 3, 4
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_split", split)
    idx.set_option("profile", 1)
    assign = np.array([[0, 2, 4, 6], [4, 1, 0, 3], [6, 5, 2, 0]], np.int32)
    tables = float_tables(rng, nq, ma, M)
    got = idx.query_scan(assign, tables.copy(), R, want_qtables=True)
    assert idx.profile()["wgq_launches"] == 1
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        assert want["rc"] == 0 and np.array_equal(got["qtables"][q], want["qtables"])
        assert heaps_equal(got["heaps"][q], (want["keys"], want["values"])), q
    # and a flat list, one query (the latency shape): 1 partition cut into `split` chunks
    flat = rand_codes(rng, 200003, M)
    i2 = pyqadc.Index(M)
    i2.add_partitions([flat])
    i2.finalize(keep)
    i2.set_option("wgq", 2)
    i2.set_option("wgq_split", split)
    t1 = float_tables(rng, 1, 1, M)
    g1 = i2.query_scan(np.zeros((1, 1), np.int32), t1.copy(), R)
    w1 = po.query_scan(M, [flat], None, keep, [0], t1[0].copy(), R)
    assert heaps_equal(g1["heaps"][0], (w1["keys"], w1["values"]))
    idx.close()
    i2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M", [16, 32])
@pytest.mark.parametrize("profiled", [0, 1])
def test_wgq_lone_small_query_shortcuts_change_nothing(pyqadc, po, M, profiled):
    """A lone small query takes three shortcuts: its input rides in the kernel arguments, its first block and the head of every
    workgroup's chunk are walked from registers loaded under the front, and the host reads its completion from the mapped result
    block (not with the profile on: then the completion event is waited for).  Repeated with different tables on the same
    slot (stale records / stale stream entries would show): heaps == oracle.  Shapes: a flat list whose first block and
    chunks are fully register-resident, one whose chunks are longer than the resident part, one shorter than the
    first block (plain walk), a sharded tail (padding-lane replays at the very end) and labels."""
    rng = np.random.default_rng(900 + M + profiled)
    keep, R = 0.05, 100                                        # (the shortest list still pre-scans more than R codes)
    for n, labelled, split, split_codes in [(100000, False, 12, 8192), (400003, True, 6, 16384), (5000, False, 12, 1024),
                                            (65536 + 7, True, 16, 4096)]:
        codes = rand_codes(rng, n, M)
        labels = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) if labelled else None
        idx = pyqadc.Index(M)
        idx.add_partitions([codes], [labels] if labelled else None)
        idx.finalize(keep)
        for k, v in (("wgq", 2), ("wgq_split", split), ("wgq_split_codes", split_codes), ("profile", profiled)):
            idx.set_option(k, v)
        for rep in range(4):
            t = float_tables(rng, 1, 1, M)
            got = idx.query_scan(np.zeros((1, 1), np.int32), t.copy(), R, want_qtables=True)
            want = po.query_scan(M, [codes], [labels] if labelled else None, keep, [0], t[0].copy(), R)
            assert want["rc"] == 0 and np.array_equal(got["qtables"][0], want["qtables"])
            assert heaps_equal(got["heaps"][0], (want["keys"], want["values"])), (n, rep)
        idx.close()


@pytest.mark.gpu
def test_wgq_resident_walk_on_ivf_shapes(pyqadc, po):
    """Small IVF batches: the register-resident start needs the first PROBED partition to hold the whole first block as
    complete vectors; queries whose first probe is short, empty or ragged take the plain walk, in the same launch as
    queries that qualify.  nq = 2 keeps the input inline (2 x 1 KiB tables would not fit with ma = 3: uploaded)."""
    M, keep, R = 16, 0.02, 100
    rng = np.random.default_rng(4242)
    sizes = [9000, 0, 8191, 8192, 30011, 100, 24001]           # 8192 codes = exactly 4096 vectors: the smallest that qualifies
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_split_codes", 2048)
    for assign in ([[3, 4, 6], [2, 4, 0]], [[1, 4, 5], [4, 3, 1]], [[0, 6, 4], [5, 0, 3]], [[6], [4]], [[3], [2]]):
        a = np.array(assign, np.int32)
        nq, ma = a.shape
        t = float_tables(rng, nq, ma, M)
        got = idx.query_scan(a, t.copy(), R)
        for q in range(nq):
            want = po.query_scan(M, parts, labels, keep, a[q], t[q].copy(), R)
            assert want["rc"] == 0 and heaps_equal(got["heaps"][q], (want["keys"], want["values"])), (assign, q)
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,nq,ma,head,big_first", [(16, 96, 8, 2, True), (16, 257, 12, 4, True), (32, 130, 6, 2, True),
                                                    (16, 70, 5, 5, True), (16, 200, 16, 3, True), (16, 96, 8, 2, False)])
def test_wgq_grouped_second_phase_matches_oracle(pyqadc, po, M, nq, ma, head, big_first):
    """Large IVF batches walk only the first `head` probes of every query in the one-workgroup-per-query kernel; the
    other (query, probe) pairs are regrouped by partition ON THE DEVICE and scanned 8 queries per pass, then every
    query's candidates are ordered back into scan order.  Labelled partitions of very different sizes (empty, shorter
    than a block, ragged), many queries per partition (groups of 8 with empty seats, partitions with several groups),
    head == ma (no second phase at all): heaps == oracle for every query.  big_first = every query's first probe is a
    long partition and later probes have larger distances (as residual tables do), so the head leaves a bound that
    holds for the rest and the grouped result is what comes back.  Without it — short head, unrelated
    random tables per probe — whole partitions fall below the head's bound, the candidate regions overflow and the
    batch is redone on the level path (tested as well)."""
    rng = np.random.default_rng(7000 + M + nq + ma)
    sizes = [int(x) for x in rng.integers(1, 3000, 37)] + [0, 15, 16, 17, 40001, 0, 129, 30000, 25013, 36000]
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
    keep, R = 0.05, 100
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    idx.set_option("wgq", 2)
    idx.set_option("wgq_group", 2)
    idx.set_option("wgq_group_head", head)
    idx.set_option("device_replay_alone_nq", 0)                 # (the grouped phase needs the device replay; see the test above)
    idx.set_option("profile", 1)
    K = len(sizes)
    big = [41, 44, 45, 46]
    assign = np.stack([rng.permutation(K)[:ma] for _ in range(nq)]).astype(np.int32)
    if big_first:
        for q in range(nq):
            b = big[q % 4]
            assign[q] = [b] + [p for p in assign[q] if p != b][:ma - 1]
    else:
        assign[:, 0] = np.where(np.isin(assign[:, 0], [37, 42]), 40, assign[:, 0])   # (first probe never empty: enough starts for qmax)
    assign[1] = assign[0]                                       # identical probe lists: shared groups all the way
    tables = float_tables(rng, nq, ma, M)
    if big_first:                                               # like real residual tables: farther centroids, larger distances
        tables = np.ascontiguousarray(tables + np.float32(0.6) * np.arange(ma, dtype=np.float32)[None, :, None])
    got = idx.query_scan(assign, tables.copy(), R, want_qtables=True)
    prof = idx.profile()
    assert prof["group_launches"] >= (1 if head < ma else 0)
    if M == 16:                                                 # (the 32x4 case holds one candidate-heavy query: > 4096 entries on any path)
        assert (prof["group_fallbacks"] > 0) == (not big_first)
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        if want["rc"] != 0:
            assert got["status"][q] == want["rc"]
            continue
        assert np.array_equal(got["qtables"][q], want["qtables"]), q
        assert heaps_equal(got["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("ma", [12, 70])
def test_ordering_pass_bucket_sort_and_its_bitonic_fallback_agree(pyqadc, po, grouped, ma):
    """The ordering pass (candidates back into scan order: assign slot, then position) sorts by (slot, position
    quarter-octave) buckets and ranks inside a bucket by counting; when the slots do not fit its scratch it takes the bitonic
    network instead: ma = 12 is the bucket sort everywhere; with ma = 70 the query kernel's own tail (scratch for 127 buckets, two
    sub-buckets per slot needed) runs the network while order_cands_kernel (1024 buckets) still sorts by buckets.  Same heaps as the
    oracle in every form, with partitions from empty to 40001 codes (positions from one bucket to all 29 quarter-octaves)."""
    rng = np.random.default_rng(515 + ma)
    M, nq, R, keep = 16, 130, 100, 0.05
    sizes = [int(x) for x in rng.integers(1, 3000, 80)] + [0, 15, 17, 40001, 129, 30000, 25013, 36000]
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    for k, v in (("wgq", 2), ("wgq_group", 2 if grouped else 0), ("wgq_group_head", 3), ("device_replay_alone_nq", 0),
                 ("profile", 1)):
        idx.set_option(k, v)
    K = len(sizes)
    big = [83, 85, 86, 87]
    assign = np.stack([rng.permutation(K)[:ma] for _ in range(nq)]).astype(np.int32)
    for q in range(nq):
        b = big[q % 4]
        assign[q] = [b] + [p_ for p_ in assign[q] if p_ != b][:ma - 1]
    tables = float_tables(rng, nq, ma, M)
    tables = np.ascontiguousarray(tables + np.float32(0.6) * np.arange(ma, dtype=np.float32)[None, :, None])
    got = idx.query_scan(assign, tables.copy(), R)
    prof = idx.profile()
    assert (prof["group_launches"] >= 1) == grouped and prof["group_fallbacks"] == 0, prof
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        assert got["status"][q] == want["rc"]
        if want["rc"] == 0:
            assert heaps_equal(got["heaps"][q], (want["keys"], want["values"])), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nq,K,dim,ma", [(300, 700, 16, 9), (257, 4100, 48, 33), (1030, 256, 128, 16), (70, 16384, 96, 64),
                                         (33, 1500, 32, 200), (20, 900, 16, 8)])
def test_coarse_assignment_of_large_batches_is_the_sequential_one(pyqadc, po, nq, K, dim, ma):
    """Batches of >= 256 queries compute the coarse distances as [16 queries] x [256 centroids] tiles and select per
    query in a second kernel; assign[] must be what the sequential host loop gives — squared L2 accumulated in
    ascending d, the ma nearest as find_k_neighbors' heaps select them (ascending distance; exact ties — duplicated
    centroids — as the heap's history and kv_binheap::sort leave them: the oracle's orc_select_k_neighbors, pinned to the
    reference's own heaps) —, a ragged last query group, K not a multiple of 256 and dimensions that are not a multiple of the
    32-wide tile."""
    rng = np.random.default_rng(nq + K)
    idx = pyqadc.Index(16)
    idx.add_partitions([rand_codes(rng, 40, 16) for _ in range(K)])
    idx.finalize(1.0)
    idx.set_pq(rng.normal(size=(16, 16, dim // 16)).astype(np.float32))
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    coarse[5] = coarse[3]
    coarse[K - 1] = coarse[K - 2]                               # exact ties, one of them in the last (partial) block
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    queries[7] = coarse[3]                                      # distance exactly 0, twice
    if K >= 900:
        # the selection's radix form (8 <= ma <= 256): many centroids tied exactly AT the ma-th distance of a query — more
        # than the 256 it sorts in LDS (the degenerate path: rounds) for query 3, a handful for query 4
        coarse[100:100 + 400] = coarse[99]
        queries[3] = coarse[99] + np.float32(0.001)
        coarse[600:600 + 5] = coarse[599]
        queries[4] = coarse[599]
    idx.set_coarse(coarse)
    got = idx.search(queries, ma, 30)["assign"]
    for q in list(range(0, nq, 17)) + [7, nq - 1]:
        d = _coarse_dists(queries[q][None, :], coarse)
        want = po.select_k_neighbors(d, ma)[0][0]
        assert np.array_equal(got[q], want), q
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nq,K,dim,ma,levels", [(300, 700, 16, 9, 3), (40, 4100, 48, 33, 2), (530, 256, 32, 16, 4), (24, 16384, 96, 64, 2),
                                                (33, 1500, 32, 200, 3), (20, 900, 16, 8, 2), (50, 600, 16, 5, 2), (9, 300, 16, 2, 2),
                                                (12, 16500, 16, 9, 2), (520, 16500, 16, 12, 3), (30, 500, 16, 256, 5), (7, 40, 16, 40, 2)])
def test_coarse_assignment_with_exact_distance_ties_is_find_k_neighbors(pyqadc, po, nq, K, dim, ma, levels):
    """Integer-valued centroids and queries: the squared distances are small integers, so most of a query's distances tie
    exactly — inside its ma nearest and across the ma-th.  The reference's choice then depends on its heap's history and on
    std::sort's order of equal keys (neighbors.cpp:18-28, 47-71; binheap.hpp:75-127); every selection kernel of the device —
    the radix form (8 <= ma <= 256), the rounds (ma < 8), the fused form (K > 16384) — must reproduce it entry for entry
    (coarse_exact_select), against the oracle's restatement that is pinned to the reference's own heaps on the CPU."""
    rng = np.random.default_rng(nq * 7 + K + ma)
    idx = pyqadc.Index(16)
    idx.add_partitions([rand_codes(rng, 40, 16) for _ in range(K)])
    idx.finalize(1.0)
    idx.set_pq(rng.normal(size=(16, 16, dim // 16)).astype(np.float32))
    coarse = rng.integers(0, levels, (K, dim)).astype(np.float32)
    queries = rng.integers(0, levels, (nq, dim)).astype(np.float32)
    idx.set_coarse(coarse)
    got = idx.search(queries, ma, 30)["assign"]
    nrule = 0
    checked = list(range(nq)) if nq <= 64 else list(range(0, nq, 13)) + [nq - 1]
    for q in checked:
        d = _coarse_dists(queries[q][None, :], coarse)
        want = po.select_k_neighbors(d, ma)[0][0]
        assert np.array_equal(got[q], want), q
        nrule += int(np.array_equal(want, np.lexsort((np.arange(K), d))[:ma]))
    assert ma == K or ma <= 2 or nrule * 4 < len(checked)       # (the plain (distance, index) rule is NOT what the heaps leave here)
    idx.close()


@pytest.mark.gpu
@path_independent
def test_ivf_encode_and_kmeans_iterations_match_numpy(pyqadc):
    """N4 through ctypes: qadc_ivf_encode_host (nearest centroid, residual, OPQ rotation, PQ encode) and
    qadc_kmeans_iterations_host: the assignment against the oracle's expansion-form coarse distances (find_k_neighbors with k = 1),
    the codes against the ORACLE's encoder (orc_pq_encode with the rotation) fed with those residuals."""
    import pyoracle as po
    rng = np.random.default_rng(99)
    M, dim, K, n = 16, 32, 50, 3000
    cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    rot = (rng.normal(size=(dim, dim)) * 0.3).astype(np.float32)
    v = rng.normal(size=(n, dim)).astype(np.float32)
    assign, codes = pyqadc.ivf_encode(cb, v, coarse=coarse, rotation=rot)
    want_assign = np.argmin(_coarse_dists(v, coarse), axis=1)   # (k = 1: the first strict minimum of the expansion distances)
    assert np.array_equal(assign, want_assign)
    res = (v - coarse[want_assign]).astype(np.float32)
    assert np.array_equal(codes, po.pq_encode(cb, res, rot))
    assert np.array_equal(pyqadc.ivf_encode(cb, v, coarse=coarse, rotation=rot, encode_form=0)[1], po.pq_encode(cb, res, rot, form=0))
    # flat database: no assignment, no residual
    _, codes_flat = pyqadc.ivf_encode(cb, v[:100])
    assert np.array_equal(codes_flat, pyqadc.pq_encode(cb, v[:100]))
    # two k-means rounds from the first K vectors: the update as the reference is COMPILED (sum * (1 / count): -ffast-math) by
    # default — checked against the reference's own loops compiled into oracle/_ref where that library is present — and as its
    # source reads (sum / count) with div_mode 0
    for mode in (1, 0):
        cen, asg = pyqadc.kmeans_iterations(v, v[:K], 2, div_mode=mode)
        c = v[:K].copy()
        for _ in range(2):
            a = np.argmin(_coarse_dists(v, c), axis=1)
            c = np.zeros_like(c)
            cnt = np.zeros(K, np.int64)
            for i in range(n):
                c[a[i]] = (c[a[i]] + v[i]).astype(np.float32)
                cnt[a[i]] += 1
            with np.errstate(invalid="ignore", divide="ignore"):
                if mode:
                    want_ref = po.reff_kmeans_update(v, a, K) if po.have_ref_float() else None
                    c = (c * (np.float32(1.0) / cnt[:, None].astype(np.float32)).astype(np.float32)).astype(np.float32)
                    assert want_ref is None or np.array_equal(want_ref, c, equal_nan=True)
                else:
                    c = (c / cnt[:, None].astype(np.float32)).astype(np.float32)
        assert np.array_equal(asg, a) and np.array_equal(cen, c, equal_nan=True), mode
