"""Soak of the lone query on ONE long partition (the sliced front, lone_front_kernel, and the walk's 32 / 64 workgroups): random list
lengths 2.5 x 10^5 ... 3 x 10^6, keep, R, table kinds; status, qmin, qmax, the int8 table and the heap against the oracle.
python tools/soak_lone_long.py <stream>;  N=<configurations> (recorded in profiles/r06_fuzz_soak_final.txt)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyqadc, pyoracle as po
po.build()
from helpers import rand_codes, float_tables, heaps_equal
base = int(sys.argv[1])
N = int(os.environ.get("N", 200))
bad = sliced = 0
t0 = time.time()
only = [int(x) for x in os.environ.get("ONLY", "").split(",") if x]
reps = int(os.environ.get("REPS", 1))
for it in (only * reps if only else range(N)):
    rng = np.random.default_rng(base * 1000003 + it)
    M = int(rng.choice([16, 32]))
    n = int(rng.integers(250000, 3000000))
    keep = float(rng.choice([0.004, 0.01, 0.02, 0.05, 0.1]))
    R = int(rng.choice([1, 2, 10, 100, 100, 257, 700]))
    labelled = bool(rng.integers(0, 2))
    parts = [rand_codes(rng, n, M)]
    labels = [rng.permutation(n).astype(np.uint32)] if labelled else None
    idx = pyqadc.Index(M); idx.add_partitions(parts, labels=labels); idx.finalize(keep)
    tb = float_tables(rng, 1, 1, M, scale=float(rng.choice([0.2, 1.0])))
    kind = int(rng.integers(0, 4))
    if kind == 1: tb = np.round(tb * 2) / 2
    if kind == 2: tb = np.where(rng.random(tb.shape) < 0.02, -np.float32(0.05) * tb, tb).astype(np.float32)
    if kind == 3: tb = (np.round(tb / 4) * 4 + 1).astype(np.float32)
    a = np.zeros((1, 1), np.int32)
    res = idx.query_scan(a, tb.copy(), R, want_qtables=True)
    want = po.query_scan(M, parts, labels, keep, a[0], tb[0].copy(), R)
    if want["rc"] != 0:
        ok = res["status"][0] == 1
    else:
        ok = (res["status"][0] == 0 and res["qmax"][0] == want["qmax"] and res["qmin"][0] == want["qmin"] and
              np.array_equal(res["qtables"][0].reshape(-1), want["qtables"].reshape(-1)) and heaps_equal(res["heaps"][0], (want["keys"], want["values"])))
    sliced += idx.profile()["lone_front_launches"]
    idx.close()
    if not ok:
        bad += 1
        print("BAD", base, it, M, n, keep, R, labelled, kind, "status", res["status"][0], "rc", want["rc"], "qmax", res["qmax"][0], want.get("qmax"),
              "qmin", res["qmin"][0], want.get("qmin"), "qt equal", want["rc"] == 0 and np.array_equal(res["qtables"][0].reshape(-1), want["qtables"].reshape(-1)), flush=True)
print("bad", bad, "of", N, "(sliced fronts: %d) in %.0f s" % (sliced, time.time() - t0))
