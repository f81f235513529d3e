"""TEST INFRASTRUCTURE — generates tests/golden/ref_encode_cases.npz: fixtures for the reference's PQ encoder
(base_pq::encode_multiple_vectors, quantizers.hpp:222-245 -> find_k_neighbors with k = 1, neighbors.cpp:30-76 ->
compute_cross_dists_blas, distances.hpp:151-215), every step the image can run produced by the REFERENCE'S OWN TEXT as
compiled here (oracle/_ref/libqadc_ref_float.so: oracle/ref_extract.sh + oracle/ref_float_harness.cpp):

    extract_subvectors                      quantizers.hpp:86-94                    qadc_reff_extract_subvectors
    ||v||^2 + ||c||^2 (the matrix sgemm     distances.hpp:151-176 / 185-208         qadc_reff_cross_norms
      receives as C, beta = 1)                (norm_4, fmanorm as compiled)
    the k = 1 selection                     neighbors.cpp:18-28 + binheap.hpp       qadc_reff_select_k_neighbors
    multiple_set_bits_4                     quantizers.hpp:49-68                    qadc_reff_pack4

and the ONE step it cannot run — cblas_sgemm(alpha = -2, beta = 1) of OpenBLAS 0.2.19, absent from this image, as is
opq's rotation sgemm (quantizers.hpp:289-301) — evaluated here in numpy float32 as one sequential dot per (vector,
centroid) in ascending d, added with a single rounding (-2 dot is exact): the restatement the oracle, the host twin and
the device share, stated as such in DESIGN.md section 6.

Run in the build container only:   make -C oracle && python oracle/gen_golden_encode.py

Per case: M, dim, codebooks [M][16][ds], vectors [n][dim], rotation [dim][dim] (opq cases), and
    norms [M][24][16]  compute_cross_dists_blas's matrix before its sgemm, per sub-quantizer, of the first 24 (ROTATED)
                       vectors (the whole matrix is what the codes were selected from; a slice keeps the file small)
    codes [n][M/2]     the codes the chain above writes
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyoracle as po  # noqa: E402

OUT = os.environ.get("QADC_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                        "ref_encode_cases.npz")


def seq_dot(x, c):
    """x [n][ds], c [k][ds] -> [n][k]: one float32 sum in ascending d."""
    dot = np.zeros((x.shape[0], c.shape[0]), np.float32)
    for d in range(x.shape[1]):
        dot = (dot + (x[:, None, d] * c[None, :, d]).astype(np.float32)).astype(np.float32)
    return dot


def seq_rotate(v, rot):
    """rotated[r] = sum_c x[c] * rotation[r][c], one float32 sum in ascending c (sgemm(NoTrans, Trans) restated)."""
    out = np.zeros_like(v)
    for r in range(rot.shape[0]):
        acc = np.zeros(v.shape[0], np.float32)
        for c in range(rot.shape[1]):
            acc = (acc + (v[:, c] * rot[r, c]).astype(np.float32)).astype(np.float32)
        out[:, r] = acc
    return out


def make_inputs(rng, M, dim, n, kind):
    ds = dim // M
    if kind == "grid":                                           # small integers: distances are exact and tie all the time
        cb = rng.integers(0, 3, (M, 16, ds)).astype(np.float32)
        v = rng.integers(0, 3, (n, dim)).astype(np.float32)
    elif kind == "offset":                                       # far from the origin, close together: the expansion's
        cb = (100.0 + 0.01 * rng.normal(size=(M, 16, ds))).astype(np.float32)   # cancellation decides, not the geometry
        v = (100.0 + 0.01 * rng.normal(size=(n, dim))).astype(np.float32)
    else:
        cb = rng.normal(size=(M, 16, ds)).astype(np.float32)
        v = rng.normal(size=(n, dim)).astype(np.float32)
    if kind in ("dup", "grid"):
        cb[:, 7] = cb[:, 2]                                      # duplicate centroids: the first one must win
        cb[1, 0] = cb[1, 15]
    if kind == "mid":                                            # vectors exactly between two centroids
        for i in range(n):
            m = int(rng.integers(0, M))
            a, b = rng.choice(16, 2, replace=False)
            v[i, m * ds:(m + 1) * ds] = (cb[m, a] + cb[m, b]) * np.float32(0.5)
    return cb, v


def ref_encode(cb, v, rot):
    M, _, ds = cb.shape
    n = v.shape[0]
    x = v if rot is None else seq_rotate(v, rot)
    norms = np.zeros((M, n, 16), np.float32)
    assign = np.zeros((n, M), np.int32)
    for m in range(M):
        sub = po.reff_extract_subvectors(x, ds, m)
        norms[m] = po.reff_cross_norms(cb[m], sub)
        dists = (norms[m] + (np.float32(-2.0) * seq_dot(sub, cb[m])).astype(np.float32)).astype(np.float32)
        assign[:, m] = po.reff_select_k_neighbors(dists, 1)[0][:, 0]
    return norms, po.reff_pack4(assign, M)


CASES = [  # cid, M, dim, n, kind, opq
    ("n16", 16, 128, 160, "normal", False), ("n32", 32, 128, 160, "normal", False), ("n32d256", 32, 256, 100, "normal", False),
    ("g16", 16, 128, 200, "grid", False), ("g32", 32, 128, 200, "grid", False),
    ("o16", 16, 128, 200, "offset", False), ("o32", 32, 128, 200, "offset", False),
    ("d16", 16, 64, 160, "dup", False), ("m16", 16, 128, 200, "mid", False), ("m32", 32, 128, 200, "mid", False),
    ("n16opq", 16, 64, 120, "normal", True), ("g32opq", 32, 128, 100, "grid", True), ("o16opq", 16, 128, 100, "offset", True),
    ("n16d480", 16, 480, 60, "normal", False), ("n16d960", 16, 960, 40, "normal", False),   # sq_dim 30 / 60: fmanorm's remainders
]


def main():
    assert po.have_ref_float(), "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(60406)
    import subprocess
    gxx = subprocess.run(["g++", "-dumpfullversion"], stdout=subprocess.PIPE).stdout.decode().strip()
    d = {"compiler": np.array("g++ %s -std=c++14 -O3 -m64 -mavx2 -mfma -mpopcnt -mbmi2 -ffast-math (oracle/Makefile REF_FLAGS)" % gxx),
         "cases": np.array([c[0] for c in CASES]), "case_meta": np.array([[c[1], c[2], c[3], int(c[5])] for c in CASES], np.int32)}
    for cid, M, dim, n, kind, opq in CASES:
        cb, v = make_inputs(rng, M, dim, n, kind)
        rot = None
        if opq:
            rot = np.linalg.qr(rng.normal(size=(dim, dim)))[0].astype(np.float32) if kind != "grid" else \
                rng.integers(-1, 2, (dim, dim)).astype(np.float32)
            d[cid + "_rotation"] = rot
        norms, codes = ref_encode(cb, v, rot)
        d[cid + "_codebooks"], d[cid + "_vectors"], d[cid + "_norms"], d[cid + "_codes"] = cb, v, norms[:, :24].copy(), codes
    np.savez_compressed(OUT, **d)
    print("wrote %s: %d cases, %d bytes" % (OUT, len(CASES), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
