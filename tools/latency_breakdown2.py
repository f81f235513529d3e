"""Where a synchronous single query's time goes (N=1e5 flat list): host submit, wait, replay."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = 16
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
idx = pyqadc.Index(M); idx.add_partition_synthetic(100000, 1); idx.finalize(0.01); idx.set_option("profile", 1)
for kv in sys.argv[1:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
q = rng.normal(size=(1, M, 1, 8)).astype(np.float32)
tb = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(1, 1, 256), np.float32)
a = np.zeros((1, 1), np.int32)
for _ in range(20): idx.query_scan(a, tb.copy(), 100)
idx.profile_reset()
ts, tc = [], []
for _ in range(300):
    t = tb.copy()
    t0 = time.perf_counter(); idx.submit(0, a, t, 100); t1 = time.perf_counter(); idx.collect(0); t2 = time.perf_counter()
    ts.append(t1 - t0); tc.append(t2 - t1)
p = idx.profile()
n = 300
print("submit %.1f us (C plan+enqueue %.1f) | collect %.1f us (C assemble %.1f, heap %.1f) | kernel %.1f us (front %.1f + scan %.1f + sort %.1f kcyc)" % (
    np.median(ts) * 1e6, p["host_plan_ms"] * 1e3 / n, np.median(tc) * 1e6, p["host_replay_ms"] * 1e3 / n, p["host_heap_ms"] * 1e3 / n,
    p["wgq_ms"] * 1e3 / max(p["wgq_launches"], 1), p["wgq_front_cycles"] / n / 1e3, p["wgq_scan_cycles"] / n / 1e3, p["wgq_sort_cycles"] / n / 1e3))
