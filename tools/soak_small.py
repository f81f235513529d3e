"""Soak of tiny configurations (partitions of 1 ... 300 codes, R 1 ... 10, 1 ... 20 queries per call, stream capacities that overflow and are
regrown, fresh index per configuration) against the oracle — the neighbourhood of the one unreproduced failure of the round-6 fuzz soak
(profiles/r06_fuzz_soak_final.txt).  python tools/soak_small.py <stream> [lib tag];  N=<configurations>; run several streams at once."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["QADC_TEST_HOOKS"] = "1"; os.environ["QADC_WGQ"] = os.environ.get("WGQ", "2"); os.environ["QADC_HEAD_LEVEL"] = "0"
import numpy as np
import pyqadc, pyoracle as po
po.build()
if len(sys.argv) > 2:
    pyqadc.LIB_PATH = os.path.join(ROOT, "quick-adc_amd", "libqadc_hip_%s.so" % sys.argv[2])
from helpers import rand_codes, float_tables, heaps_equal
base_seed = int(sys.argv[1])
N = int(os.environ.get("N", 20000))
bad = 0
t0 = time.time()
for it in range(N):
    seed = base_seed * 10000000 + it
    rng = np.random.default_rng(seed)
    M = int(rng.choice([16, 32]))
    nparts = int(rng.integers(1, 4))
    sizes = [int(rng.choice([1, 2, 15, 16, 17, 33, 100, 300])) for _ in range(nparts)]
    parts = [rand_codes(rng, n, M) for n in sizes]
    R = int(rng.choice([1, 2, 3, 10]))
    keep = float(rng.choice([0.01, 0.05, 0.3]))
    nq = int(rng.integers(1, 21))
    ma = int(rng.integers(1, nparts + 1))
    assign = np.stack([rng.permutation(nparts)[:ma] for _ in range(nq)]).astype(np.int32)
    idx = pyqadc.Index(M); idx.add_partitions(parts, labels=None); idx.finalize(keep)
    idx.set_option("wgq_capacity", int(rng.choice([64, 4096]))); idx.set_option("wgq_split", int(rng.choice([1, 3, 8])))
    idx.set_option("device_replay_nq", int(rng.choice([0, 1]))); idx.set_option("device_replay_alone_nq", int(rng.choice([0, 512])))
    tables = float_tables(rng, nq, ma, M, scale=float(rng.choice([0.2, 1.0])))
    if rng.integers(0, 2): tables = np.round(tables * 2) / 2
    if rng.integers(0, 4) == 0: tables = np.where(rng.random(tables.shape) < 0.01, -np.float32(0.02) * tables, tables).astype(np.float32)
    res = idx.query_scan(assign, tables.copy(), R)
    for q in range(nq):
        want = po.query_scan(M, parts, None, keep, assign[q], tables[q].copy(), R)
        ok = (res["status"][q] == 1) if want["rc"] != 0 else (res["status"][q] == 0 and heaps_equal(res["heaps"][q], (want["keys"], want["values"])))
        if not ok:
            bad += 1
            print("BAD seed", seed, "q", q, "M", M, "sizes", sizes, "R", R, "keep", keep, "nq", nq, "ma", ma, "assign", assign[q], "status", res["status"][q], "want rc", want["rc"], "qmax", res["qmax"][q], flush=True)
            break
    idx.close()
print("bad", bad, "of", N, "in %.0f s" % (time.time() - t0))
