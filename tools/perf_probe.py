"""Quick single-GPU probe: int8 scan kernel rate on a synthetic flat partition (not the bench)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc  # noqa: E402

M = int(os.environ.get("M", 16))
N = int(float(os.environ.get("N", 1e9)))
NQ = int(os.environ.get("NQ", 4))
REPS = int(os.environ.get("REPS", 3))
rng = np.random.default_rng(0)
idx = pyqadc.Index(M)
t = time.time()
idx.add_partition_synthetic(N, 0x5EED0001)
idx.finalize(0.01)
print("generated %d codes in %.2fs" % (N, time.time() - t), flush=True)
idx.set_option("profile", 1)
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    idx.set_option(k, float(v))
qt = rng.integers(0, 14, (NQ, 1, M, 16)).astype(np.int8)
assign = np.zeros((NQ, 1), np.int32)
idx.scan_i8(assign, qt, 100)  # warmup
for r in range(REPS):
    idx.profile_reset()
    t = time.time()
    heaps = idx.scan_i8(assign, qt, 100)
    dt = time.time() - t
    p = idx.profile()
    gbs = p["scan_codes"] * (M // 2) / (p["scan_ms"] * 1e-3) / 1e9
    print("rep %d: wall %.2f ms/query | kernel %.3f ms/query over %d launches -> %.1f GB/s (%.1f%% of 8 TB/s) | "
          "cands %d replay %.2f ms | heap max %d" % (r, dt * 1e3 / NQ, p["scan_ms"] / NQ, p["scan_launches"], gbs,
                                                     gbs / 80, p["candidates"], p["host_replay_ms"],
                                                     heaps[0][1].max()), flush=True)
