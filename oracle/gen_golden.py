"""TEST INFRASTRUCTURE — generates tests/golden/ref_scan_cases.npz from the REFERENCE BUILD
(oracle/_ref/libqadc_ref.so = the reference's own binheap.hpp / simd_layout.hpp / simd_scan.hpp
compiled from /root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle && python oracle/gen_golden.py

A fixture is data: inputs (codes or the seed of the counter-based generator, int8 tables, labels,
R) and the reference's outputs (interleaved bytes, final heap arrays, sort_keys order).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyoracle as po  # noqa: E402

OUT = os.environ.get("QADC_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                        "ref_scan_cases.npz")


def synth_codes(n, M, seed):
    cs = M // 2
    nwords = (n * cs + 7) // 8
    return po.fill_codes(0, nwords, seed)[:n * cs].reshape(n, cs).copy()


def main():
    assert po.have_ref(), "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20171)
    d = {}
    cases = []
    # -- single-partition, explicit codes (small): edge sizes, ties, saturation, R > N ---------
    for M in (16, 32):
        for n in (1, 15, 16, 17, 37, 50, 1000):
            for tmax, R in ((2, 10), (20, 100), (127, 100)):
                cid = "c%03d" % len(cases)
                codes = rng.integers(0, 256, (n, M // 2), dtype=np.uint8)
                qt = rng.integers(0, tmax + 1, (1, M, 16)).astype(np.int8)
                keys, vals, skeys = po.ref_scan(M, [codes], None, qt, R, want_sorted=True)
                d[cid + "_codes0"] = codes
                d[cid + "_qt"] = qt
                d[cid + "_keys"] = keys
                d[cid + "_vals"] = vals
                d[cid + "_sorted"] = skeys
                if n <= 50:
                    d[cid + "_inter0"] = po.ref_interleave(codes)
                cases.append((cid, M, R, 1, 0, 0))
    # -- multi-partition with labels, one shared heap (IVF shape), sizes with n % 16 != 0 -------
    for M in (16, 32):
        for tmax, R in ((4, 100), (25, 100), (60, 20)):
            cid = "c%03d" % len(cases)
            sizes = [333, 17, 2048, 1, 999]
            parts = [rng.integers(0, 256, (s, M // 2), dtype=np.uint8) for s in sizes]
            perm = rng.permutation(sum(sizes)).astype(np.uint32) + 7
            labels, o = [], 0
            for s in sizes:
                labels.append(perm[o:o + s].copy())
                o += s
            qt = rng.integers(0, tmax + 1, (len(sizes), M, 16)).astype(np.int8)
            keys, vals, skeys = po.ref_scan(M, parts, labels, qt, R, want_sorted=True)
            for i, (p, l) in enumerate(zip(parts, labels)):
                d["%s_codes%d" % (cid, i)] = p
                d["%s_labels%d" % (cid, i)] = l
            d[cid + "_qt"] = qt
            d[cid + "_keys"] = keys
            d[cid + "_vals"] = vals
            d[cid + "_sorted"] = skeys
            cases.append((cid, M, R, len(sizes), 1, 0))
    # -- larger flat partitions from the counter-based generator (only the seed is stored) -------
    for M, n, seed, tmax, R in ((16, 100003, 11, 12, 100), (32, 65537, 12, 9, 100), (16, 1000000, 13, 10, 100),
                               (16, 250000, 14, 127, 50)):
        cid = "c%03d" % len(cases)
        codes = synth_codes(n, M, seed)
        qt = rng.integers(0, tmax + 1, (1, M, 16)).astype(np.int8)
        keys, vals, skeys = po.ref_scan(M, [codes], None, qt, R, want_sorted=True)
        d[cid + "_synth"] = np.array([n, seed], np.int64)
        d[cid + "_qt"] = qt
        d[cid + "_keys"] = keys
        d[cid + "_vals"] = vals
        d[cid + "_sorted"] = skeys
        cases.append((cid, M, R, 1, 0, 1))
    d["cases"] = np.array([c[0] for c in cases])
    d["case_meta"] = np.array([c[1:] for c in cases], np.int64)  # M, R, nparts, has_labels, synthetic
    # -- heap push sequences (kv_binheap<unsigned,int8_t> / <unsigned,float>) ---------------------
    for i, (n, R, vmax) in enumerate(((5, 10, 3), (200, 10, 3), (5000, 100, 50), (3000, 1, 127), (4000, 64, 1))):
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        vals = rng.integers(0, vmax + 1, n).astype(np.int8)
        ok, ov = po.ref_heap_replay_i8(keys, vals, R)
        d["h%d_in_keys" % i], d["h%d_in_vals" % i], d["h%d_R" % i] = keys, vals, np.array(R)
        d["h%d_keys" % i], d["h%d_vals" % i] = ok, ov
        fv = (rng.integers(0, 40, n) / np.float32(8)).astype(np.float32)  # many exact float ties
        fk, fvv = po.ref_heap_replay_f32(keys, fv, R)
        d["f%d_in_vals" % i], d["f%d_keys" % i], d["f%d_vals" % i] = fv, fk, fvv
    d["n_heap_cases"] = np.array(5)
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(cases), "scan cases")


if __name__ == "__main__":
    main()
