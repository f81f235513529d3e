"""Host-side timing of submit/collect to check that the two-slot pipeline overlaps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
N = int(float(os.environ.get("N", 1e9))); NQ = int(os.environ.get("NQ", 8)); M = 16
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
q = rng.normal(size=(NQ, M, 1, 8)).astype(np.float32)
tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, 256), np.float32)
assign = np.zeros((NQ, 1), np.int32)
idx.submit(0, assign, tables.copy(), 100); idx.collect(0)
t0 = time.perf_counter(); pend = None
for s in range(int(os.environ.get("STEPS", 6))):
    a = time.perf_counter(); idx.submit(s % 2, assign, tables.copy(), 100); b = time.perf_counter()
    if pend is not None:
        idx.collect(pend)
    c = time.perf_counter(); pend = s % 2
    print("step %d: submit %.2f ms, collect %.2f ms, t=%.2f" % (s, (b - a) * 1e3, (c - b) * 1e3, (c - t0) * 1e3))
idx.collect(pend); print("total %.2f ms" % ((time.perf_counter() - t0) * 1e3))
