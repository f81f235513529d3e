"""Blast radius of the parity-UNPINNED float half, measured on the CPU oracle only.

The reference's QuantizerMAX (db_query_4.cpp:37-71) and its pre-scan sum (query_common.hpp:59-90) are compiled
with -ffast-math: the as-compiled quantizer multiplies by a reciprocal (quant_mode 1) where the source divides
(quant_mode 0), and qmax may move by an ulp if the compiler reassociates the pre-scan adds.  Neither can be pinned
here (the translation unit does not build in this image).  This sweep answers: how often would either difference
change what a query RETURNS?  For >= 2000 queries on real-encoded data (flat and IVF, keep 1 % and 0.2 %) it
compares the final heap of the default path (quant_mode 1, qmax as computed) with
  (a) quant_mode 0 at the same qmax,  (b) qmax + 1 ulp,  (c) qmax - 1 ulp  (both quant_mode 1)
and reports the fraction of queries whose returned KEY SET differs, whose heap ARRAYS differ, and the fraction of
int8 table entries that differ.  Output: JSON (committed as profiles/r02_quantizer_sensitivity.json).

    python tools/sensitivity_sweep.py [nqueries] > profiles/r02_quantizer_sensitivity.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

M, DIM, R = 16, 128, 100
DS = DIM // M


def make_data(rng, n, nclusters):
    centres = rng.normal(size=(nclusters, DIM)).astype(np.float32) * 3
    base = centres[rng.integers(0, nclusters, n)] + rng.normal(size=(n, DIM)).astype(np.float32)
    return centres, base


def encode(cb, x):
    """Plain PQ encode: nearest centroid per sub-quantizer (first minimum), two per byte, even one in the low nibble."""
    n = x.shape[0]
    ids = np.zeros((n, M), np.int32)
    for m in range(M):
        d = ((x[:, None, m * DS:(m + 1) * DS] - cb[m][None]) ** 2).sum(-1)
        ids[:, m] = d.argmin(1)
    return po.pack4(ids, M)


def tables_for(cb, resid):
    return np.ascontiguousarray(((resid.reshape(-1, M, 1, DS) - cb[None]) ** 2).sum(-1, dtype=np.float32).reshape(-1, M * 16))


def run_config(name, parts, labels, keep, queries_assign_tables):
    stats = dict(queries=0, skipped_qmax_too_high=0)
    for k in ("mode0", "qmax_plus_ulp", "qmax_minus_ulp"):
        stats[k] = dict(key_set_differs=0, heap_arrays_differ=0, table_entries_differ=0, table_entries=0)
    for assign, tables in queries_assign_tables:
        base = po.query_scan(M, parts, labels, keep, assign, tables, R, quant_mode=1)   # clamps `tables` in place
        if base["rc"] != 0:
            stats["skipped_qmax_too_high"] += 1
            continue
        stats["queries"] += 1
        qmin, qmax = np.float32(base["qmin"]), np.float32(base["qmax"])
        probed = [parts[p] for p in assign]
        plab = None if labels is None else [labels[p] for p in assign]
        live = [i for i, p in enumerate(probed) if len(p)]
        variants = {"mode0": (qmax, 0), "qmax_plus_ulp": (np.nextafter(qmax, np.float32(np.inf)), 1),
                    "qmax_minus_ulp": (np.nextafter(qmax, np.float32(-np.inf)), 1)}
        for k, (qm, mode) in variants.items():
            qt = po.quantize_tables(tables, qmin, qm, mode).reshape(len(assign), M, 16)
            keys, vals = po.scan_i8(M, [probed[i] for i in live], None if plab is None else [plab[i] for i in live],
                                    qt[live], R)
            st = stats[k]
            st["key_set_differs"] += int(set(keys.tolist()) != set(base["keys"].tolist()))
            st["heap_arrays_differ"] += int(not (np.array_equal(keys, base["keys"]) and np.array_equal(vals, base["values"])))
            st["table_entries_differ"] += int((qt != base["qtables"]).sum())
            st["table_entries"] += qt.size
    q = max(stats["queries"], 1)
    out = dict(config=name, keep=keep, queries=stats["queries"], skipped_qmax_too_high=stats["skipped_qmax_too_high"])
    for k in ("mode0", "qmax_plus_ulp", "qmax_minus_ulp"):
        st = stats[k]
        out[k] = dict(frac_queries_key_set_differs=st["key_set_differs"] / q,
                      frac_queries_heap_arrays_differ=st["heap_arrays_differ"] / q,
                      frac_table_entries_differ=st["table_entries_differ"] / max(st["table_entries"], 1))
    return out


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(20261003)
    t0 = time.time()
    results = []
    # ---- flat: 100 000 real-encoded codes ----
    n_flat = 100000
    centres, base = make_data(rng, n_flat, 400)
    cb = np.stack([base[rng.integers(0, n_flat, 16), m * DS:(m + 1) * DS] for m in range(M)]).astype(np.float32)
    codes = encode(cb, base)
    queries = centres[rng.integers(0, len(centres), nq)] + rng.normal(size=(nq, DIM)).astype(np.float32)
    tb = tables_for(cb, queries)
    for keep in (0.01, 0.002):
        results.append(run_config("flat, %d real-encoded 16x4 codes, R=%d" % (n_flat, R), [codes], None, keep,
                                  (([0], tb[q:q + 1].copy()) for q in range(nq))))
    # ---- IVF: 1 000 000 codes, K = 64 coarse centroids (sampled vectors), residual encoding, ma = 8, labels ----
    n_ivf, K, ma = 1000000, 64, 8
    centres, base = make_data(rng, n_ivf, 2000)
    coarse = base[rng.integers(0, n_ivf, K)].copy()
    d2 = (base ** 2).sum(1)[:, None] - 2 * base @ coarse.T + (coarse ** 2).sum(1)[None]
    owner = d2.argmin(1)
    resid = base - coarse[owner]
    cb = np.stack([resid[rng.integers(0, n_ivf, 16), m * DS:(m + 1) * DS] for m in range(M)]).astype(np.float32)
    codes = encode(cb, resid)
    perm = rng.permutation(n_ivf).astype(np.uint32)
    parts = [np.ascontiguousarray(codes[owner == k]) for k in range(K)]
    labels = [np.ascontiguousarray(perm[owner == k]) for k in range(K)]
    queries = centres[rng.integers(0, len(centres), nq)] + rng.normal(size=(nq, DIM)).astype(np.float32)
    qd = (queries ** 2).sum(1)[:, None] - 2 * queries @ coarse.T + (coarse ** 2).sum(1)[None]
    assign = np.argsort(qd, axis=1, kind="stable")[:, :ma].astype(np.int32)

    def ivf_queries():
        for q in range(nq):
            r = queries[q][None] - coarse[assign[q]]
            yield assign[q], tables_for(cb, r)

    for keep in (0.01, 0.002):
        results.append(run_config("IVF, %d real-encoded 16x4 codes, K=%d, ma=%d, labels, R=%d" % (n_ivf, K, ma, R),
                                  parts, labels, keep, ivf_queries()))
    print(json.dumps({"what": __doc__.split("\n\n")[1].replace("\n", " "), "seconds": round(time.time() - t0, 1),
                      "results": results}, indent=1))


if __name__ == "__main__":
    main()
