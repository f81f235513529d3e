"""GPU: the lone synchronous query and small batches — one query spread over several workgroups of the query kernel
(scan_query_kernel, MULTI): the first block in one step (per-epoch bounds, workgroup 0 writing its candidates straight into the
stream), workgroup 0 without a chunk, chunk candidates in LDS, the select's fitted-digit pass.  Targeted at the seams the
randomised sweep (test_gpu_fuzz.py) only hits by chance: list lengths around the first block, R around the number of starts,
tie-heavy tables, capacities that overflow, many starts, chunks that cross partitions.  Oracle = scanner_4::query_scan restated
(db_query_4.cpp:245-309), bit-exact."""
import numpy as np
import pytest

from helpers import float_tables, heaps_equal, path_independent, rand_codes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


def check(idx, po, M, parts, labels, keep, assign, tables, R):
    res = idx.query_scan(assign, tables.copy(), R)
    for q in range(assign.shape[0]):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R)
        if want["rc"] != 0:
            assert res["status"][q] == 1, q
            continue
        assert res["status"][q] == 0, q
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), q


def make(pyqadc, M, sizes, keep, labelled, seed, **opts):
    rng = np.random.default_rng(seed)
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes] if labelled else None
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels=labels)
    idx.finalize(keep)
    idx.set_option("wgq", 1)                                     # (this file is about the query kernel: whatever the suite's hooks say)
    for k, v in opts.items():
        idx.set_option(k, v)
    return rng, parts, labels, idx


@path_independent                                                # (one path only — the query kernel's — set by hand above)
@pytest.mark.parametrize("M", [16, 32])
def test_list_lengths_around_the_first_block(pyqadc, po, M):
    """The first block is 4096 vectors (8192 codes at 16x4, 4096 at 32x4) of the first probed partition.  A list of exactly that
    length keeps its last code — the one with padding-lane replays — out of the one-step form; one code more is in."""
    nc = 4096 * (32 // M)
    for n in (nc - 1, nc, nc + 1, nc + 15, nc + 16, nc + 17, nc + 64 * (32 // M) - 1, 2 * nc, 2 * nc + 2049, 5 * nc + 3):
        for labelled in (False, True):
            rng, parts, labels, idx = make(pyqadc, M, [n], 0.01, labelled, 10 + n + labelled, wgq_split_codes=1024)
            tb = float_tables(rng, 2, 1, M)
            check(idx, po, M, parts, labels, 0.01, np.zeros((2, 1), np.int32), tb, 100)
            idx.close()


@path_independent
@pytest.mark.parametrize("M", [16, 32])
def test_r_against_the_number_of_starts_and_tie_heavy_tables(pyqadc, po, M):
    """R = 1, R just below / at / above the number of pre-scanned starts (above: qmax = FLT_MAX, the reference's exit path),
    R beyond the wave replay; tables on a coarse grid (the fitted-digit select then meets buckets of hundreds of equal keys and
    falls back to the fixed passes; the int8 sums tie massively)."""
    n, keep = 60000, 0.01
    rng, parts, labels, idx = make(pyqadc, M, [n], keep, True, 77 + M)
    starts = int(n * keep)
    a = np.zeros((3, 1), np.int32)
    for R in (1, 2, starts - 1, starts, starts + 1, 257, 1000):
        tb = float_tables(rng, 3, 1, M, scale=0.5)
        tb[1] = np.round(tb[1] * 2) / 2                          # tie-heavy
        tb[2] = np.round(tb[2] / 4) * 4 + 1                      # a handful of distinct values
        check(idx, po, M, parts, labels, keep, a, tb, R)
    idx.close()


@path_independent
def test_capacities_that_overflow(pyqadc, po):
    """A stream capacity below what workgroup 0 writes straight from the first block (regrown, re-run), a candidate capacity below
    what a chunk emits (the batch falls back to the level path), and both at once."""
    M, n = 16, 120000
    for opts in (dict(wgq_capacity=64), dict(wgq_cand_cap=8), dict(wgq_capacity=16, wgq_cand_cap=4), dict(wgq_capacity=300)):
        rng, parts, labels, idx = make(pyqadc, M, [n], 0.01, False, 5, **opts)
        tb = float_tables(rng, 4, 1, M)
        tb[3] = 1.0                                              # constant tables: every code is a candidate
        check(idx, po, M, parts, labels, 0.01, np.zeros((4, 1), np.int32), tb, 100)
        assert idx.profile()["regrows"] >= 1, opts
        idx.close()


@path_independent
@pytest.mark.parametrize("M", [16, 32])
def test_many_starts_and_chunks_across_partitions(pyqadc, po, M):
    """keep = 30 %: thousands of starts per query (the threshold cut of the select; beyond the LDS budget at 32x4 the global
    scratch), a first partition just long enough for the one-step form, chunks that run across the later partitions (one of
    them empty, one a single code), 1 ... 9 queries per call."""
    sizes = [4096 * (32 // M) + 77, 0, 30011, 1, 52000, 9000]
    rng, parts, labels, idx = make(pyqadc, M, sizes, 0.3, True, 31 + M)
    for nq in (1, 2, 5, 9):
        # partition 0 first (the one-step form needs the first probed partition long enough), the others in every query's own order
        assign = np.stack([np.concatenate([[0], 1 + rng.permutation(5)]) for _ in range(nq)]).astype(np.int32)
        tb = float_tables(rng, nq, 6, M, scale=0.3)
        check(idx, po, M, parts, labels, 0.3, assign, tb, 100)
    idx.close()


@path_independent
def test_workgroups_per_query(pyqadc, po):
    """2 ... 64 workgroups per query on the same list: the answer does not depend on how the scan order is cut."""
    M, n = 16, 150001
    rng, parts, labels, idx = make(pyqadc, M, [n], 0.01, False, 9)
    tb = float_tables(rng, 1, 1, M)
    a = np.zeros((1, 1), np.int32)
    for split, codes in ((2, 1024), (3, 8192), (7, 1024), (12, 8192), (32, 2048), (64, 1024)):
        idx.set_option("wgq_split", split)
        idx.set_option("wgq_split_codes", codes)
        check(idx, po, M, parts, labels, 0.01, a, tb, 100)
    idx.close()


@path_independent
@pytest.mark.parametrize("M", [16, 32])
def test_int8_tables_in(pyqadc, po, M):
    """The caller's int8 tables (qadc_scan_i8: no front in the kernel; the one-step first block stages the table itself), one
    and two queries per call, list lengths around the first block, small and saturating table entries."""
    from helpers import rand_qtables
    nc = 4096 * (32 // M)
    for n in (nc, nc + 1, 3 * nc + 5, 120001):
        rng, parts, labels, idx = make(pyqadc, M, [n], 0.01, n % 2 == 1, 3 + n, wgq_split_codes=1024)
        for nq in (1, 2):
            for tmax, R in ((3, 10), (12, 100), (127, 100), (40, 1)):
                qt = rand_qtables(rng, (nq, 1), M, tmax)
                got = idx.scan_i8(np.zeros((nq, 1), np.int32), qt, R)
                for q in range(nq):
                    want = po.scan_i8(M, parts, labels, qt[q], R)
                    assert heaps_equal(got[q], want), (M, n, nq, tmax, R, q)
        idx.close()


@path_independent
def test_lone_query_on_a_long_flat_list(pyqadc, po):
    """One or two queries on 1.5 M codes: by default the query kernel's 32 workgroups per query (chunks of ~47 K codes that refresh
    their bounds epoch by epoch), not the level path's chain of launches; three queries take the level path.  Same answers."""
    M, n = 16, 1500003
    rng, parts, labels, idx = make(pyqadc, M, [n], 0.01, True, 123)
    for nq in (1, 2, 3):
        tb = float_tables(rng, nq, 1, M)
        check(idx, po, M, parts, labels, 0.01, np.zeros((nq, 1), np.int32), tb, 100)
    idx.close()


@path_independent
@pytest.mark.parametrize("M", [16, 32])
def test_sliced_front_of_a_lone_query(pyqadc, po, M):
    """One query, one long partition: the front runs in a launch of its own, sliced over workgroups (lone_front_kernel: every slice's
    R smallest pre-scan values, the last workgroup takes the R-th smallest of their union, qmin / clamp / QuantizerMAX), and the walk
    launch behind it takes the int8 table and {flags, qmin, qmax} from it.  qmax, qmin, the int8 table, the clamp of the caller's
    tables and the heap against the oracle; R from 1 up to where S x R outgrows the last workgroup's budget (the walk's own front
    then); slices that end ragged; tie-heavy and partly negative tables; fewer starts than R is the reference's exit path."""
    rng = np.random.default_rng(900 + M)
    for n, keep in ((1000003, 0.01), (300000, 0.05), (2500000, 0.0041), (900000, 0.3)):
        parts = [rand_codes(rng, n, M)]
        idx = pyqadc.Index(M)
        idx.add_partitions(parts, labels=None)
        idx.finalize(keep)
        idx.set_option("wgq", 1)
        a = np.zeros((1, 1), np.int32)
        for R, kind in ((100, "plain"), (1, "plain"), (100, "ties"), (257, "negative"), (1000, "plain"), (4000, "plain")):
            tb = float_tables(rng, 1, 1, M, scale=0.5)
            if kind == "ties":
                tb = np.round(tb * 2) / 2
            if kind == "negative":
                tb = np.where(rng.random(tb.shape) < 0.02, -np.float32(0.05) * tb, tb).astype(np.float32)
            t_in = tb.copy()
            res = idx.query_scan(a, t_in, R, want_qtables=True)
            want = po.query_scan(M, parts, None, keep, a[0], tb[0].copy(), R)
            if want["rc"] != 0:
                assert res["status"][0] == 1, (n, keep, R, kind)
                continue
            assert res["status"][0] == 0, (n, keep, R, kind)
            assert res["qmax"][0] == want["qmax"] and res["qmin"][0] == want["qmin"], (n, keep, R, kind)
            assert np.array_equal(res["qtables"][0].reshape(-1), want["qtables"].reshape(-1)), (n, keep, R, kind)
            assert np.array_equal(t_in, np.where(tb < 0, np.float32(0), tb)), (n, keep, R, kind)      # clamped in place like the reference
            assert heaps_equal(res["heaps"][0], (want["keys"], want["values"])), (n, keep, R, kind)
        sliced = idx.profile()["lone_front_launches"]
        idx.close()
        assert (sliced >= 3) == (keep < 0.3), (n, keep, sliced)      # (keep = 0.3 on 9 x 10^5 codes: 66 slices, one too many — the walk's own front)
