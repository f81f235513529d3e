"""Where a synchronous single query spends its time (host enqueue vs wait)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = 16; rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
N = int(float(os.environ.get("N", 1e6)))
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
q = rng.normal(size=(1, M, 1, 8)).astype(np.float32)
tb = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(1, 1, 256), np.float32)
a = np.zeros((1, 1), np.int32)
for _ in range(5): idx.query_scan(a, tb.copy(), 100)
idx.profile_reset(); n = 100
ts = tc = 0.0
for _ in range(n):
    t0 = time.perf_counter(); idx.submit(0, a, tb.copy(), 100); t1 = time.perf_counter(); idx.collect(0); t2 = time.perf_counter()
    ts += t1 - t0; tc += t2 - t1
p = idx.profile()
print("N=%d: submit %.1f us (library plan+enqueue %.1f us), collect %.1f us (asm %.1f heap %.1f us)" % (
    N, ts / n * 1e6, p["host_plan_ms"] / n * 1e3, tc / n * 1e6, p["host_replay_ms"] / n * 1e3, p["host_heap_ms"] / n * 1e3))
