"""Blast radius of the parity-UNPINNED float half, measured on the CPU oracle only.

The reference's QuantizerMAX (db_query_4.cpp:37-71) and its pre-scan sum (query_common.hpp:59-90) are compiled
with -ffast-math: the as-compiled quantizer multiplies by a reciprocal (quant_mode 1) where the source divides
(quant_mode 0), and qmax may move by an ulp if the compiler reassociates the pre-scan adds.  Neither can be pinned
here (the translation unit does not build in this image).  This sweep answers: how often would either difference
change what a query RETURNS?  For >= 2000 queries on real-encoded data (flat and IVF, keep 1 % and 0.2 %) it
compares the final heap of the default path (quant_mode 1, qmax as computed) with
  (a) quant_mode 0 at the same qmax,  (b) qmax + 1 ulp,  (c) qmax - 1 ulp  (both quant_mode 1)
and reports the fraction of queries whose returned KEY SET differs, whose heap ARRAYS differ, and the fraction of
int8 table entries that differ.

Round 3 adds the DOMINANT noise source: the reference's float tables themselves come from FMA kernels (fmanorm,
distances.hpp:60-92) or OpenBLAS (distances.hpp:151-183), so EVERY table entry may differ from this repo's sequential
sums by an ulp or a few, and a reassociated 16-term pre-scan sum can move qmax by more than one ulp.  Variants
  (d) every float table entry moved by a random integer in [-1, +1] ulp,  (e) ... in [-4, +4] ulp
      (the WHOLE query is re-run on the perturbed tables: new qmin, new pre-scan, new qmax, new int8 tables),
  (f) qmax + 4 ulp,  (g) qmax - 4 ulp
and a configuration at the BASELINE configs[2] proportions (nprobe 32 of many partitions).
Output: JSON (committed as profiles/r03_table_noise_sensitivity.json).

    python tools/sensitivity_sweep.py [nqueries] > profiles/r03_table_noise_sensitivity.json
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


VARIANTS = ("mode0", "qmax_plus_ulp", "qmax_minus_ulp", "qmax_plus_4ulp", "qmax_minus_4ulp", "tables_1ulp", "tables_4ulp")


def step_ulps(x, k):
    """float32 array moved by k[i] ulps (k integer array; positive floats, so the bit pattern is monotone)."""
    bits = x.view(np.int32).astype(np.int64) + k
    return np.maximum(bits, 0).astype(np.int32).view(np.float32)


def run_config(name, parts, labels, keep, queries_assign_tables, noise_rng):
    stats = dict(queries=0, skipped_qmax_too_high=0)
    for k in VARIANTS:
        stats[k] = dict(key_set_differs=0, heap_arrays_differ=0, table_entries_differ=0, table_entries=0, keys_changed=0)
    for assign, tables in queries_assign_tables:
        pristine = tables.copy()
        base = po.query_scan(M, parts, labels, keep, assign, tables, R, quant_mode=1)   # clamps `tables` in place
        if base["rc"] != 0:
            stats["skipped_qmax_too_high"] += 1
            continue
        stats["queries"] += 1
        qmin, qmax = np.float32(base["qmin"]), np.float32(base["qmax"])
        probed = [parts[p] for p in assign]
        plab = None if labels is None else [labels[p] for p in assign]
        live = [i for i, p in enumerate(probed) if len(p)]
        up, down = np.float32(np.inf), np.float32(-np.inf)

        def moved(q, n, to):
            for _ in range(n):
                q = np.nextafter(q, to)
            return q

        variants = {"mode0": (qmax, 0), "qmax_plus_ulp": (moved(qmax, 1, up), 1), "qmax_minus_ulp": (moved(qmax, 1, down), 1),
                    "qmax_plus_4ulp": (moved(qmax, 4, up), 1), "qmax_minus_4ulp": (moved(qmax, 4, down), 1)}

        def tally(k, keys, vals, qt):
            st = stats[k]
            diff = set(keys.tolist()) ^ set(base["keys"].tolist())
            st["key_set_differs"] += int(bool(diff))
            st["keys_changed"] += len(diff) // 2
            st["heap_arrays_differ"] += int(not (np.array_equal(keys, base["keys"]) and np.array_equal(vals, base["values"])))
            st["table_entries_differ"] += int((qt != base["qtables"]).sum())
            st["table_entries"] += qt.size

        for k, (qm, mode) in variants.items():
            qt = po.quantize_tables(tables, qmin, qm, mode).reshape(len(assign), M, 16)
            keys, vals = po.scan_i8(M, [probed[i] for i in live], None if plab is None else [plab[i] for i in live],
                                    qt[live], R)
            tally(k, keys, vals, qt)
        for k, amp in (("tables_1ulp", 1), ("tables_4ulp", 4)):
            noisy = np.ascontiguousarray(step_ulps(np.maximum(pristine, np.float32(1e-30)),
                                                   noise_rng.integers(-amp, amp + 1, pristine.shape)), np.float32)
            noisy[pristine <= 0] = pristine[pristine <= 0]
            r2 = po.query_scan(M, parts, labels, keep, assign, noisy, R, quant_mode=1)
            if r2["rc"] != 0:
                continue
            tally(k, r2["keys"], r2["values"], r2["qtables"])
    q = max(stats["queries"], 1)
    out = dict(config=name, keep=keep, queries=stats["queries"], skipped_qmax_too_high=stats["skipped_qmax_too_high"])
    for k in VARIANTS:
        st = stats[k]
        out[k] = dict(frac_queries_key_set_differs=st["key_set_differs"] / q,
                      frac_queries_heap_arrays_differ=st["heap_arrays_differ"] / q,
                      keys_changed_per_differing_query=st["keys_changed"] / max(st["key_set_differs"], 1),
                      frac_table_entries_differ=st["table_entries_differ"] / max(st["table_entries"], 1))
    return out


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    small = len(sys.argv) > 2 and sys.argv[2] == "small"        # (the CPU test suite: same code, ten times less data)
    rng = np.random.default_rng(20261003)
    t0 = time.time()
    results = []
    # ---- flat: 100 000 real-encoded codes ----
    n_flat = 20000 if small else 100000
    centres, base = make_data(rng, n_flat, 400)
    cb = np.stack([base[rng.integers(0, n_flat, 16), m * DS:(m + 1) * DS] for m in range(M)]).astype(np.float32)
    codes = encode(cb, base)
    queries = centres[rng.integers(0, len(centres), nq)] + rng.normal(size=(nq, DIM)).astype(np.float32)
    tb = tables_for(cb, queries)
    for keep in (0.01, 0.002):
        results.append(run_config("flat, %d real-encoded 16x4 codes, R=%d" % (n_flat, R), [codes], None, keep,
                                  (([0], tb[q:q + 1].copy()) for q in range(nq)), np.random.default_rng(77)))
    # ---- IVF: 1 000 000 codes, K = 64 coarse centroids (sampled vectors), residual encoding, ma = 8, labels; and the
    # BASELINE configs[2] proportions: 2 000 000 codes, K = 1024, ma = 32 (32 of many short partitions) ----
    for n_ivf, K, ma, nclusters, nq_cfg in (((100000, 64, 8, 500, nq), (200000, 256, 32, 800, nq)) if small else
                                            ((1000000, 64, 8, 2000, nq), (2000000, 1024, 32, 4000, max(200, nq // 2)))):
        centres, base = make_data(rng, n_ivf, nclusters)
        coarse = base[rng.integers(0, n_ivf, K)].copy()
        owner = np.zeros(n_ivf, np.int64)
        for lo in range(0, n_ivf, 100000):                          # (chunked: n x K distances do not fit at K = 1024)
            b = base[lo:lo + 100000]
            owner[lo:lo + 100000] = ((b ** 2).sum(1)[:, None] - 2 * b @ coarse.T + (coarse ** 2).sum(1)[None]).argmin(1)
        resid = base - coarse[owner]
        cb = np.stack([resid[rng.integers(0, n_ivf, 16), m * DS:(m + 1) * DS] for m in range(M)]).astype(np.float32)
        codes = np.concatenate([encode(cb, resid[lo:lo + 200000]) for lo in range(0, n_ivf, 200000)])
        perm = rng.permutation(n_ivf).astype(np.uint32)
        order = np.argsort(owner, kind="stable")
        bounds = np.searchsorted(owner[order], np.arange(K + 1))
        parts = [np.ascontiguousarray(codes[order[bounds[k]:bounds[k + 1]]]) for k in range(K)]
        labels = [np.ascontiguousarray(perm[order[bounds[k]:bounds[k + 1]]]) for k in range(K)]
        queries = centres[rng.integers(0, len(centres), nq_cfg)] + rng.normal(size=(nq_cfg, DIM)).astype(np.float32)
        qd = (queries ** 2).sum(1)[:, None] - 2 * queries @ coarse.T + (coarse ** 2).sum(1)[None]
        assign = np.argsort(qd, axis=1, kind="stable")[:, :ma].astype(np.int32)

        def ivf_queries():
            for q in range(nq_cfg):
                r = queries[q][None] - coarse[assign[q]]
                yield assign[q], tables_for(cb, r)

        for keep in (0.01, 0.002):
            results.append(run_config("IVF, %d real-encoded 16x4 codes, K=%d, ma=%d, labels, R=%d" % (n_ivf, K, ma, R),
                                      parts, labels, keep, ivf_queries(), np.random.default_rng(78)))
    print(json.dumps({"what": __doc__.split("\n\n")[1].replace("\n", " "), "seconds": round(time.time() - t0, 1),
                      "results": results}, indent=1))


if __name__ == "__main__":
    main()
