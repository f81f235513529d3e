"""CPU-only.  How much does the ORDER in which an sgemm adds its products matter for what this repository restates as one
sequential dot?  The reference's cblas_sgemm is OpenBLAS 0.2.19's, which is not in this image; numpy's
float32 matmul calls the OpenBLAS that numpy bundles (another version, other kernels: NOT the reference's either) — a second,
real summation order to compare with.  For the encoder (find_k_neighbors with k = 1 on the expansion distances) and for the
distance tables, on data shaped like bench.py's real-encode legs:

    python3 tools/sgemm_order_sensitivity.py  ->  profiles/r06_sgemm_order_sensitivity.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def ulps(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def ivf_queries(rng):
    """End to end on the CPU oracle: an IVF database (K = 64 cells, 200 K clustered 128-d vectors, residuals PQ 16x4 encoded by the
    oracle's encoder), nprobe 8, R = 100, keep 1 %: every query scanned twice — with the expansion tables as this repository builds
    them (sequential dot) and with numpy's sgemm in the product's place — same assign[], same codes."""
    M, dim, K, ma, R, keep, n, nq = 16, 128, 64, 8, 100, 0.01, 200000, 400
    ds = dim // M
    centres = 3 * rng.normal(size=(500, dim)).astype(np.float32)
    base = (centres[rng.integers(0, 500, n)] + rng.normal(size=(n, dim))).astype(np.float32)
    queries = (centres[rng.integers(0, 500, nq)] + rng.normal(size=(nq, dim))).astype(np.float32)
    coarse = base[rng.integers(0, n, K)].copy()
    d2 = ((base ** 2).sum(1)[:, None] - 2 * base @ coarse.T + (coarse ** 2).sum(1)[None]).astype(np.float32)
    asg = d2.argmin(1)
    res = (base - coarse[asg]).astype(np.float32)
    cb = np.stack([res[rng.integers(0, n, 16), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
    codes = po.pq_encode(cb, res)
    parts = [np.ascontiguousarray(codes[asg == k]) for k in range(K)]
    labels = [np.nonzero(asg == k)[0].astype(np.uint32) for k in range(K)]
    cn = np.stack([po.cross_dists(cb[m], np.zeros((1, ds), np.float32), with_product=False)[0] for m in range(M)])   # ||c||^2 as compiled
    differ_sets = differ_heaps = differ_tables = exits = 0
    for q in range(nq):
        dq = ((queries[q][None, :] - coarse) ** 2).sum(1)
        assign = np.argsort(dq, kind="stable")[:ma].astype(np.int32)
        r = (queries[q][None, :] - coarse[assign]).astype(np.float32)                                  # [ma][dim]
        t_seq = po.tables_expansion(cb, r)                                                              # [ma][M*16]
        t_blas = np.zeros_like(t_seq)
        for m in range(M):
            sub = np.ascontiguousarray(r[:, m * ds:(m + 1) * ds])
            norms = po.cross_dists(cb[m], sub, with_product=False)
            t_blas[:, m * 16:(m + 1) * 16] = (norms + np.float32(-2.0) * (sub @ cb[m].T)).astype(np.float32)
        a = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(t_seq), R)
        b = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(t_blas), R)
        if a["rc"] or b["rc"]:
            exits += 1
            continue
        differ_tables += int(not np.array_equal(a["qtables"], b["qtables"]))
        differ_heaps += int(not (np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["values"], b["values"])))
        differ_sets += int(set(a["keys"].tolist()) != set(b["keys"].tolist()))
    return {"what": ivf_queries.__doc__.split("\n")[0], "queries": nq, "exit_path": exits,
            "queries_whose_int8_tables_differ": differ_tables, "queries_whose_heap_arrays_differ": differ_heaps,
            "queries_whose_key_SET_differs": differ_sets}


def coarse_assignment(rng):
    """assign[] of 2000 queries over K = 4096 centroids (128-d, clustered), nprobe 32, through find_k_neighbors' selection (the oracle's
    pinned restatement): the reference's FORM with a real BLAS in it (||q||^2 + ||c||^2 as compiled, then numpy's sgemm) against (a) this
    library's coarse distances — the same form with one sequential dot — and (b) the direct form sum (q - c)^2 it used until round 6."""
    K, dim, nq, ma = 4096, 128, 2000, 32
    centres = 3 * rng.normal(size=(1000, dim)).astype(np.float32)
    coarse = (centres[rng.integers(0, 1000, K)] + rng.normal(size=(K, dim))).astype(np.float32)
    queries = (centres[rng.integers(0, 1000, nq)] + rng.normal(size=(nq, dim))).astype(np.float32)
    direct = np.zeros((nq, K), np.float32)
    for d in range(dim):
        t = (queries[:, d:d + 1] - coarse[None, :, d]).astype(np.float32)
        direct = (direct + (t * t).astype(np.float32)).astype(np.float32)
    norms = np.concatenate([po.cross_dists(coarse[k0:k0 + 256], queries, with_product=False) for k0 in range(0, K, 256)], 1)   # BLOCK_NEIGHS = 256
    blas = (norms + np.float32(-2.0) * (queries @ coarse.T)).astype(np.float32)
    seq = np.concatenate([po.cross_dists(coarse[k0:k0 + 256], queries) for k0 in range(0, K, 256)], 1)
    a_blas = po.select_k_neighbors(blas, ma)[0]
    res = {"what": coarse_assignment.__doc__.split("\n")[0], "queries": nq, "K": K, "nprobe": ma}
    for name, dmat in (("expansion_form_sequential_dot(the library)", seq), ("direct_form(until round 6)", direct)):
        a = po.select_k_neighbors(dmat, ma)[0]
        res[name] = {"queries_with_the_same_assign_array": int((a == a_blas).all(1).sum()),
                     "queries_with_the_same_probed_SET": int(sum(set(x.tolist()) == set(y.tolist()) for x, y in zip(a, a_blas)))}
    return res


def main():
    rng = np.random.default_rng(606)
    out = {"what": __doc__.split("\n\n")[0], "numpy": np.__version__, "cases": []}
    for M, dim, kind in ((16, 128, "clustered"), (32, 128, "clustered"), (32, 96, "clustered"), (16, 128, "normal")):
        ds, n = dim // M, 200000
        if kind == "clustered":
            centres = 3 * rng.normal(size=(2000, dim)).astype(np.float32)
            v = (centres[rng.integers(0, 2000, n)] + rng.normal(size=(n, dim))).astype(np.float32)
        else:
            v = rng.normal(size=(n, dim)).astype(np.float32)
        cb = np.stack([v[rng.integers(0, n, 16), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)   # sampled sub-vectors
        diff_assign = 0
        worst, tiny_abs, entries, differing = [], 0.0, 0, 0
        for m in range(M):
            sub = np.ascontiguousarray(v[:, m * ds:(m + 1) * ds])
            norms = po.cross_dists(cb[m], sub, with_product=False)                 # pinned half
            seq = po.cross_dists(cb[m], sub)                                       # + the sequential dot (what the library computes)
            blas = (norms + np.float32(-2.0) * (sub @ cb[m].T)).astype(np.float32)  # + numpy's sgemm
            a_seq = po.select_k_neighbors(seq, 1)[0][:, 0]
            a_blas = po.select_k_neighbors(blas, 1)[0][:, 0]
            diff_assign += int((a_seq != a_blas).sum())
            big = np.abs(seq) > np.float32(1e-3) * np.median(np.abs(seq))          # (ulps mean nothing across a sign change at ~0)
            worst.append(int(ulps(seq[big], blas[big]).max()))
            tiny_abs = max(tiny_abs, float(np.abs(seq[~big] - blas[~big]).max()) if (~big).any() else 0.0)
            entries += seq.size
            differing += int((seq != blas).sum())
        case = {"M": M, "dim": dim, "sq_dim": ds, "data": kind, "vectors": n,
                "sub_quantizer_assignments": n * M, "assignments_that_differ": diff_assign,
                "fraction": diff_assign / (n * M), "distance_entries": entries, "entries_that_differ_in_any_bit": differing,
                "max_ulp_difference(entries above 1e-3 of the median distance)": max(worst),
                "max_abs_difference_of_the_near_zero_entries": tiny_abs}
        out["cases"].append(case)
        print(case, flush=True)
    out["ivf_queries"] = ivf_queries(rng)
    out["coarse_assignment"] = coarse_assignment(rng)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_sgemm_order_sensitivity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
