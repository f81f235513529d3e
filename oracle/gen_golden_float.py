"""TEST INFRASTRUCTURE — generates tests/golden/ref_query_scan_cases.npz: fixtures for the float -> int8 boundary of
scanner_4::query_scan (db_query_4.cpp:230-309), every output produced by the REFERENCE'S OWN CODE as compiled here:

    oracle/_ref/libqadc_ref_float.so   scanner_4 (whole), QuantizerMAX, scan_4 — line ranges of the reference's files
                                       (oracle/ref_extract.sh, oracle/ref_float_harness.cpp)
    oracle/_ref/libqadc_ref.so         scan_avx_4 + kv_binheap (oracle/ref_harness.cpp)

Run in the build container only:   make -C oracle && python oracle/gen_golden_float.py

Per case the file holds the inputs (the database it runs on — row-major partitions, or the seed of the counter-based
generator for the larger flat ones —, labels, keep, R, assign, float tables BEFORE the in-place clamp) and the
reference's results:
    exit     exit status of query_scan in a child process (1 = "Max quantization bound too high", db_query_4.cpp:271-274)
    starts   scanner_4::starts_sizes of the probed partitions (prepare_database, db_query_4.cpp:125-126)
    qmax     tmp_bh.max() after scanner_4::query_scan_start (230-242, 259)
    qmin     min of all tables clamped at 0 (258-263) — a comparison, no arithmetic: evaluated here in numpy
    qt       QuantizerMAX<int8_t>(qmin, qmax).quantize_tables of the CLAMPED tables (277-284)        [exit == 0]
    keys / vals / sorted   the int8 heap after the WHOLE scanner_4::query_scan, and sort_keys of it   [exit == 0]
and, as a cross-check made at generation time, the heap from scan_avx_4 fed with `qt` equals keys / vals.
Four small cases also carry per-code arrays: `fcand` = scan_4's float sum of every start code (read out of a float heap
with room for all of them) and `cand` = scan_avx_4's int8 value of every code of the first probed partition (read out
of an int8 heap with room for all; codes the scan never pushes have cand = 127).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyoracle as po  # noqa: E402

OUT = os.environ.get("QADC_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                        "ref_query_scan_cases.npz")


def dist_tables(rng, ma, M, scale=1.0, negatives=0.0, levels=0):
    """[ma][M*16] float32 distance-table-like values; `negatives` = fraction of slightly negative entries (what the BLAS
    expansion form produces); `levels` > 0 = that many distinct values only (ties everywhere)."""
    d = 8
    q = rng.normal(size=(ma, M, 1, d)).astype(np.float32)
    c = rng.normal(size=(1, M, 16, d)).astype(np.float32)
    t = ((q - c) ** 2).sum(-1).astype(np.float32) * np.float32(scale)
    if levels:
        t = (np.floor(t / t.max() * levels) * np.float32(scale / levels)).astype(np.float32)
    if negatives:
        neg = rng.random(t.shape) < negatives
        t = np.where(neg, -np.float32(0.03 * scale) * rng.random(t.shape).astype(np.float32), t).astype(np.float32)
    return np.ascontiguousarray(t.reshape(ma, M * 16), np.float32)


def synth_codes(n, M, seed):
    cs = M // 2
    return po.fill_codes(0, (n * cs + 7) // 8, seed)[:n * cs].reshape(n, cs).copy()


def main():
    assert po.have_ref() and po.have_ref_float(), "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20175)
    d, cases = {}, []

    dbs = []

    def store_db(parts, labels, synth=None):
        """-> database id; partitions are stored once however many cases query them (synth = (n, seed): only that)."""
        dbi = len(dbs)
        if synth is not None:
            d["db%d_synth" % dbi] = np.array(synth, np.int64)
        else:
            for i, p in enumerate(parts):
                d["db%d_codes%d" % (dbi, i)] = np.ascontiguousarray(p, np.uint8)
        if labels is not None:
            for i, l in enumerate(labels):
                d["db%d_labels%d" % (dbi, i)] = np.ascontiguousarray(l, np.uint32)
        dbs.append((len(parts), int(labels is not None), int(synth is not None)))
        return dbi

    def add(M, parts, labels, keep, R, assign, tables, per_code=False, dbi=None, synth=None):
        cid = "q%02d" % len(cases)
        if dbi is None:
            dbi = store_db(parts, labels, synth)
        assign = np.asarray(assign, np.int32)
        ma = len(assign)
        tables = np.ascontiguousarray(tables, np.float32).reshape(ma, M * 16)
        sc = po.RefScanner4(M, parts, labels, keep)
        starts, psizes, has_labels = sc.sizes()
        assert has_labels == (labels is not None)
        fk, fv = sc.query_start(assign, tables, R)
        qmax = fv[0]
        qmin = np.float32(0) if tables.min() < 0 else tables.min()
        clamped = np.where(tables < 0, np.float32(0), tables).astype(np.float32)
        code = sc.try_query(assign, tables, R)
        assert code == (1 if float(qmax) > 1e30 else 0), (cid, code, qmax)
        d[cid + "_assign"] = assign
        d[cid + "_tables"] = tables.copy()
        d[cid + "_exit"] = np.array(code)
        d[cid + "_starts"] = starts[assign].copy()
        d[cid + "_qmax"] = np.array(qmax, np.float32)
        d[cid + "_qmin"] = np.array(qmin, np.float32)
        if code == 0:
            qt = po.reff_quantize_tables(clamped.reshape(ma, M, 16), float(qmin), float(qmax))
            tb = tables.copy()
            keys, vals, skeys = sc.query_scan(assign, tb, R, want_sorted=True)
            assert np.array_equal(tb, clamped), cid                    # the in-place clamp (262-269)
            # cross-check: the integer half alone (scan_avx_4 + kv_binheap of libqadc_ref.so) on those int8 tables
            live = [a for a in range(ma) if parts[assign[a]].shape[0] > 0]
            k2, v2 = po.ref_scan(M, [parts[assign[a]] for a in live],
                                 None if labels is None else [labels[assign[a]] for a in live], qt[live], R)
            assert np.array_equal(k2, keys) and np.array_equal(v2, vals), cid
            d[cid + "_qt"], d[cid + "_keys"], d[cid + "_vals"], d[cid + "_sorted"] = qt, keys, vals, skeys
            if per_code:
                p0 = parts[assign[0]]
                n0, s0 = p0.shape[0], int(starts[assign[0]])
                pk, pv = po.reff_scan4_start(M, [p0[:s0]], None, tables[:1], s0 + 2)       # room for all: nothing evicted
                fc = np.full(s0, np.nan, np.float32)
                fc[pk[1:]] = pv[1:]                                    # (entry 0 is the (0, FLT_MAX) seed, never moved: it is the max)
                assert pk[0] == 0 and pv[0] == np.finfo(np.float32).max and not np.isnan(fc).any(), cid
                d[cid + "_fcand"] = fc
                ck, cv = po.ref_scan(M, [p0], None, qt[:1], n0 + 32)   # room for all, incl. the padding lanes' replays
                cand = np.full(n0, 127, np.int8)
                cand[ck[1:]] = cv[1:]
                d[cid + "_cand"] = cand
        cases.append((cid, M, R, dbi, int(labels is not None), int(per_code), float(keep)))
        sc.close()

    def codes(n, M):
        return rng.integers(0, 256, (n, M // 2), dtype=np.uint8)

    R = 100
    # ---- flat: scales, negatives, ties; per-code arrays on the small ones ------------------------------------------
    for M in (16, 32):
        add(M, [codes(1800, M)], None, 0.08, R, [0], dist_tables(rng, 1, M), per_code=True)
        add(M, [codes(2101, M)], None, 0.06, R, [0], dist_tables(rng, 1, M, negatives=0.03), per_code=True)
        add(M, [synth_codes(40003, M, 21 + M)], None, 0.01, R, [0], dist_tables(rng, 1, M, scale=1000.0), synth=(40003, 21 + M))
        add(M, [synth_codes(30000, M, 22 + M)], None, 0.01, R, [0], dist_tables(rng, 1, M, scale=1e-3, negatives=0.01),
            synth=(30000, 22 + M))
        add(M, [synth_codes(9000, M, 23 + M)], None, 0.05, R, [0], dist_tables(rng, 1, M, levels=6), synth=(9000, 23 + M))
    # ---- start sizes 1 ... R + 50 around the "heap not full -> FLT_MAX -> exit" edge (db_query_4.cpp:271-274) -----
    for s in (1, R - 1, R, R + 1, R + 50):
        n = 1000
        add(16, [codes(n, 16)], None, (s + 0.5) / n, R, [0], dist_tables(rng, 1, 16))
    add(16, [codes(900, 16)], None, 0.5, 1000, [0], dist_tables(rng, 1, 16))          # R > N
    add(16, [codes(5000, 16)], None, 0.05, 10, [0], dist_tables(rng, 1, 16, negatives=0.02))   # small R
    # ---- IVF: labels, one (qmin, qmax) for all probes, ragged sizes, an empty probed partition ----------------------
    for M, keep in ((16, 0.05), (32, 0.04), (16, 0.02)):
        sizes = [3333, 17, 0, 2048, 1, 4999, 777, 1600]
        parts = [codes(s, M) for s in sizes]
        perm = rng.permutation(sum(sizes)).astype(np.uint32) + 11
        labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
        dbi = store_db(parts, labels)
        for assign in ([5, 0, 3, 7], [2, 1, 4, 6, 0], [7, 6, 5, 4, 3, 2, 1, 0]):
            add(M, parts, labels, keep, R, assign, dist_tables(rng, len(assign), M, negatives=0.02 if M == 16 else 0.0), dbi=dbi)
    d["cases"] = np.array([c[0] for c in cases])
    d["case_meta"] = np.array([c[1:6] for c in cases], np.int64)          # M, R, database id, has_labels, per_code
    d["db_meta"] = np.array(dbs, np.int64)                                # nparts, has_labels, synthetic
    d["case_keep"] = np.array([c[6] for c in cases], np.float32)
    np.savez_compressed(OUT, **d)
    nexit = sum(int(d[c[0] + "_exit"]) for c in cases)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(cases), "query_scan cases,", nexit, "of them exit(1)")


if __name__ == "__main__":
    main()
