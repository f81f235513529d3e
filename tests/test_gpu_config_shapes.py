"""IVF at the shape of BASELINE.json's configurations (C3 = IVF 16x4, nprobe 32, K = 4096;
C5 = IVF 32x4 on 96-d vectors (sq_dim 3, which the reference's own dispatch cannot do, distances.cpp:15-121),
nprobe 64): queries go in through qadc_search (coarse assignment, residuals, BLAS-expansion tables, pre-scan,
quantizer, scan, heap — all on the GPU) and every sampled query must end in the oracle's heap arrays.  The oracle
gets the assignments and float tables from a numpy restatement of the same sequential float loops."""
import numpy as np
import pytest

from helpers import heaps_equal, rand_codes


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


def _coarse_dists(x, c):
    """The coarse distances of the query row(s) x [1 or n][dim] to the centroids c [K][dim] as find_k_neighbors gets them
    (compute_cross_dists_blas, distances.hpp:151-183): the ORACLE's orc_cross_dists — (||x||^2 + ||c||^2) with the norms as the
    reference compiles them, then -2 x.c as one sequential dot.  Returns [K] for one row, [n][K] for several."""
    import pyoracle
    d = pyoracle.cross_dists(c, np.ascontiguousarray(x, np.float32).reshape(-1, c.shape[1]))
    return d[0] if d.shape[0] == 1 else d


def _seq_expansion(x, c):
    """The BLAS-expansion table form of one residual (x [M][1][ds], c = codebooks [M][16][ds]) -> [M][16]: the ORACLE's
    orc_tables_expansion (norm half pinned to the reference as compiled, the sgemm's product one sequential dot)."""
    import pyoracle
    return pyoracle.tables_expansion(c, x.reshape(-1)).reshape(c.shape[0], 16)


def _build(pyqadc, rng, M, K, N, dim, keep):
    sizes = rng.multinomial(N, np.ones(K) / K).astype(np.int64)
    sizes[rng.integers(0, K, 3)] = 0                             # a few empty partitions, as real indexes have
    codes_all = rand_codes(rng, int(sizes.sum()), M)
    perm = rng.permutation(int(sizes.sum())).astype(np.uint32)   # labels != positions
    cuts = np.cumsum(sizes)[:-1]
    parts, labels = np.split(codes_all, cuts), np.split(perm, cuts)
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels)
    idx.finalize(keep)
    cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    idx.set_pq(cb)
    idx.set_coarse(coarse)
    return idx, parts, labels, cb, coarse, sizes


def _check(po, res, sample, queries, parts, labels, cb, coarse, M, ma, keep, R):
    K, dim = coarse.shape
    ds = dim // M
    for q in sample:
        dist = _coarse_dists(queries[q][None, :], coarse)
        assign = np.lexsort((np.arange(K), dist))[:ma].astype(np.int32)
        assert np.array_equal(res["assign"][q], assign), q
        resid = (queries[q][None, :] - coarse[assign]).astype(np.float32)
        tables = np.stack([_seq_expansion(resid[a].reshape(M, 1, ds), cb) for a in range(ma)])     # ma > 1: expansion form
        want = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(tables.reshape(ma, M * 16)), R)
        assert want["rc"] == res["status"][q] == 0, q
        sz = res["sizes"][q]
        assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), (want["keys"], want["values"])), q


@pytest.mark.gpu
def test_c3_shape_ivf_16x4_k4096_nprobe32(pyqadc, po):
    """BASELINE configs[2] shape at 1.2e7 codes: K = 4096 ragged labelled partitions, nprobe = 32, 128-d, R = 100,
    keep = 1 %, a 1024-query batch (the batch size bench.py's `ivf` leg pipelines); 48 sampled queries vs the oracle."""
    rng = np.random.default_rng(4096)
    M, K, N, dim, ma, R, keep, nq = 16, 4096, 12_000_000, 128, 32, 100, 0.01, 1024
    idx, parts, labels, cb, coarse, sizes = _build(pyqadc, rng, M, K, N, dim, keep)
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    res = idx.search(queries, ma, R)
    assert int(res["status"].sum()) == 0
    _check(po, res, rng.choice(nq, 48, replace=False), queries, parts, labels, cb, coarse, M, ma, keep, R)
    idx.close()


@pytest.mark.gpu
def test_c5_shape_ivf_32x4_dim96_nprobe64(pyqadc, po):
    """BASELINE configs[4] shape on one GPU's worth of data: 32x4 codes (16 B), 96-d vectors => sq_dim 3 (the generic
    sub-vector loops; the reference's table dispatch has no such case), K = 1024, nprobe = 64, R = 100, 4e6 codes;
    a 513-query batch, 32 sampled queries (the last one always) vs the oracle."""
    rng = np.random.default_rng(96)
    M, K, N, dim, ma, R, keep, nq = 32, 1024, 4_000_000, 96, 64, 100, 0.01, 513   # (513: a ragged last group of the 4-queries-per-workgroup coarse kernel)
    idx, parts, labels, cb, coarse, sizes = _build(pyqadc, rng, M, K, N, dim, keep)
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    res = idx.search(queries, ma, R)
    assert int(res["status"].sum()) == 0
    _check(po, res, list(rng.choice(nq - 1, 31, replace=False)) + [nq - 1], queries, parts, labels, cb, coarse, M, ma, keep, R)
    idx.close()


@pytest.mark.gpu
def test_c3_full_size_the_bench_list_against_oracle(pyqadc, po):
    """BASELINE configs[2] at FULL size — the very list bench.py's `ivf` leg times: 10^8 synthetic 16x4 codes in K = 4096
    partitions (multinomial sizes, partition p = generator stream 1000 + p), nprobe 32, a 1024-query batch through
    qadc_search (head + partition-major second phase).  The oracle only needs the probed partitions of the sampled
    queries, which the CPU generator reproduces: 12 sampled queries must end in the oracle's heaps."""
    M, K, MA, nq, dim, N, R, keep = 16, 4096, 32, 1024, 128, 100_000_000, 100, 0.01
    rng = np.random.default_rng(0)
    sizes = rng.multinomial(N, np.ones(K) / K)
    idx = pyqadc.Index(M)
    for p in range(K):
        idx.add_partition_synthetic(int(sizes[p]), 1000 + p)
    idx.finalize(keep)
    cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    idx.set_pq(cb)
    idx.set_coarse(coarse)
    idx.set_option("profile", 1)
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    res = idx.search(queries, MA, R)
    prof = idx.profile()
    import os
    if os.environ.get("QADC_WGQ") != "0":                       # (the fixture also runs this test on the level-structured path)
        assert prof["group_launches"] >= 1 and prof["group_fallbacks"] == 0   # the path bench.py measures, not a fallback
    assert int(res["status"].sum()) == 0
    ds = dim // M
    for q in rng.choice(nq, 12, replace=False):
        dist = _coarse_dists(queries[q][None, :], coarse)
        assign = np.lexsort((np.arange(K), dist))[:MA].astype(np.int32)
        assert np.array_equal(res["assign"][q], assign), q
        resid = (queries[q][None, :] - coarse[assign]).astype(np.float32)
        tables = np.stack([_seq_expansion(resid[a].reshape(M, 1, ds), cb) for a in range(MA)])
        probed = [po.fill_codes(0, int(sizes[p]), 1000 + int(p)).reshape(-1, M // 2) for p in assign]
        want = po.query_scan(M, probed, None, keep, np.arange(MA, dtype=np.int32),
                             np.ascontiguousarray(tables.reshape(MA, M * 16)), R)
        assert want["rc"] == 0
        sz = res["sizes"][q]
        assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), (want["keys"], want["values"])), q
    # The integer half of ALL 1024 queries against the REFERENCE BUILD itself (oracle/_ref: the reference's scan_avx_4<16>,
    # simd_scan.hpp:125-187, and interleave_partition_4, simd_layout.hpp:55-65): the device's own int8 tables, the probed
    # partitions regenerated on the CPU and laid out by the reference, one shared heap per query in assign[] order.
    # Budget: ~4 s to generate + interleave the 4096 partitions, ~1 s for 1024 x 0.78 M codes through the AVX2 kernel.
    if po.have_ref():
        inter = {}
        qt_all = idx.slot_qtables(0, 0, nq, MA)
        for q in range(nq):
            assign = res["assign"][q]
            for p in assign:
                if int(p) not in inter:
                    inter[int(p)] = po.ref_interleave(po.fill_codes(0, int(sizes[p]), 1000 + int(p)).reshape(-1, M // 2))
            live = [a for a in range(MA) if sizes[assign[a]] > 0]
            want = po.ref_scan_interleaved(M, [inter[int(assign[a])] for a in live], [int(sizes[assign[a]]) for a in live], None,
                                           qt_all[q][live], R)
            sz = res["sizes"][q]
            assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), want), q
    idx.close()


@pytest.mark.gpu
def test_c5_full_size_1e9_codes_32x4_nprobe64_on_one_gpu(pyqadc, po):
    """BASELINE configs[4] at FULL size on ONE GPU (16 GB of codes): 10^9 synthetic 32x4 codes in K = 16384 partitions,
    96-d vectors (sq_dim 3), nprobe 64, a 1024-query batch through qadc_search; 6 sampled queries (3.9 M probed codes
    each, regenerated on the CPU) must end in the oracle's heaps."""
    M, K, MA, nq, dim, N, R, keep = 32, 16384, 64, 1024, 96, 1_000_000_000, 100, 0.01
    rng = np.random.default_rng(5)
    sizes = rng.multinomial(N, np.ones(K) / K)
    idx = pyqadc.Index(M)
    for p in range(K):
        idx.add_partition_synthetic(int(sizes[p]), 7000 + p)
    idx.finalize(keep)
    cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    idx.set_pq(cb)
    idx.set_coarse(coarse)
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    res = idx.search(queries, MA, R)
    assert int(res["status"].sum()) == 0
    ds = dim // M
    for q in rng.choice(nq, 6, replace=False):
        dist = _coarse_dists(queries[q][None, :], coarse)
        assign = np.lexsort((np.arange(K), dist))[:MA].astype(np.int32)
        assert np.array_equal(res["assign"][q], assign), q
        resid = (queries[q][None, :] - coarse[assign]).astype(np.float32)
        tables = np.stack([_seq_expansion(resid[a].reshape(M, 1, ds), cb) for a in range(MA)])
        # a 32x4 code is 16 bytes = two words of the generator stream
        probed = [po.fill_codes(0, 2 * int(sizes[p]), 7000 + int(p)).reshape(-1, M // 2) for p in assign]
        want = po.query_scan(M, probed, None, keep, np.arange(MA, dtype=np.int32),
                             np.ascontiguousarray(tables.reshape(MA, M * 16)), R)
        assert want["rc"] == 0
        sz = res["sizes"][q]
        assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), (want["keys"], want["values"])), q
    # integer half against the reference build (scan_avx_4<32> on the device's int8 tables), 48 sampled queries of
    # 3.9 M probed codes each; budget ~25 s (the probed partitions are regenerated per query: 64 x 1 MB)
    if po.have_ref():
        for q in rng.choice(nq, 48, replace=False):
            assign = res["assign"][q]
            qt = idx.slot_qtables(0, int(q), 1, MA)[0]
            live = [a for a in range(MA) if sizes[assign[a]] > 0]
            inter = [po.ref_interleave(po.fill_codes(0, 2 * int(sizes[assign[a]]), 7000 + int(assign[a])).reshape(-1, M // 2)) for a in live]
            want = po.ref_scan_interleaved(M, inter, [int(sizes[assign[a]]) for a in live], None, qt[live], R)
            sz = res["sizes"][q]
            assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), want), q
    idx.close()
