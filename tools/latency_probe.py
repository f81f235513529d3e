"""Single-query (nq=1, synchronous) latency of qadc_query_scan for small databases."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = 16
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
for N in (100000, 1000000, 10000000):
    idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
    for opt in sys.argv[1:]:
        k, v = opt.split("="); idx.set_option(k, float(v))
    q = rng.normal(size=(1, M, 1, 8)).astype(np.float32)
    tb = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(1, 1, 256), np.float32)
    a = np.zeros((1, 1), np.int32)
    for _ in range(5): idx.query_scan(a, tb.copy(), 100)
    t = time.perf_counter(); n = 50
    for _ in range(n): idx.query_scan(a, tb.copy(), 100)
    dt = (time.perf_counter() - t) / n
    print("N=%9d: %.1f us per synchronous single query (%.2e codes/s)" % (N, dt * 1e6, N / dt), flush=True)
    idx.close()
