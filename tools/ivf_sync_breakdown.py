"""Synchronous qadc_search at the C3 shape: where a small batch's time goes (host plan / wait / heap, kernel)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M, K, MA, dim, N, R = 16, 4096, 32, 128, 100_000_000, 100
rng = np.random.default_rng(0)
sizes = rng.multinomial(N, np.ones(K) / K)
idx = pyqadc.Index(M)
for p in range(K):
    idx.add_partition_synthetic(int(sizes[p]), 1000 + p)
idx.finalize(0.01)
idx.set_pq(rng.normal(size=(M, 16, dim // M)).astype(np.float32))
idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
idx.set_option("profile", 1)
for kv in sys.argv[1:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
for nq in (1, 8, 32, 64, 128, 256):
    q = rng.normal(size=(nq, dim)).astype(np.float32)
    for _ in range(3):
        idx.search(q, MA, R)
    idx.profile_reset()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        idx.search(q, MA, R)
    dt = (time.perf_counter() - t0) / n * 1e3
    p = idx.profile()
    print("nq %4d: %.3f ms per call | host plan %.3f  stream assembly %.3f  heap %.3f | kernel (events) %.3f ms | stream entries/query %.0f" % (
        nq, dt, p["host_plan_ms"] / n, p["host_replay_ms"] / n, p["host_heap_ms"] / n, p["wgq_ms"] / max(p["wgq_launches"], 1),
        p["candidates"] / n / nq))
