"""Flat list of N codes (default 1e7 = BASELINE configs[1]), 32 queries per step, three steps in flight: step time and the
library's own host / GPU accounting per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M, N, NQ, R = 16, int(float(os.environ.get("N", 1e7))), int(os.environ.get("NQ", 32)), 100
idx = pyqadc.Index(M)
idx.add_partition_synthetic(N, 0x5EED0001)
idx.finalize(0.01)
idx.set_option("profile", 1)
for kv in sys.argv[1:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
rng = np.random.default_rng(1)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
def tabs():
    q = rng.normal(size=(NQ, M, 1, 8)).astype(np.float32)
    return np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, 256), np.float32)
pool = [tabs() for _ in range(4)]
a = np.zeros((NQ, 1), np.int32)
def run(k):
    pend = []
    for s in range(k):
        idx.submit(s % 3, a, pool[s % 4].copy(), R)
        pend.append(s % 3)
        if len(pend) == 3:
            idx.collect(pend.pop(0))
    while pend:
        idx.collect(pend.pop(0))
run(10)
idx.profile_reset()
t0 = time.perf_counter()
K = 300
run(K)
dt = (time.perf_counter() - t0) / K * 1e3
p = idx.profile()
print("N %.0e x %d queries: %.3f ms/step = %.3e codes/s | per step: host plan %.3f  assembly %.3f  heap %.3f | GPU streaming launches %.3f ms (%d), start %.3f | candidates/query %.0f" % (
    N, NQ, dt, N * NQ / dt * 1e3, p["host_plan_ms"] / K, p["host_replay_ms"] / K, p["host_heap_ms"] / K, p["scan_ms"] / K, p["scan_launches"] / K,
    p["start_ms"] / K, p["candidates"] / K / NQ))
