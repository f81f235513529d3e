"""A/B sweep of scan-kernel variants in ONE process, interleaved rounds (guide rule 24)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = int(os.environ.get("M", 16)); N = int(float(os.environ.get("N", 1e9))); NQ = int(os.environ.get("NQ", 4))
ROUNDS = int(os.environ.get("ROUNDS", 3))
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 0x5EED0001); idx.finalize(0.01); idx.set_option("profile", 1)
rng = np.random.default_rng(0)
qt = rng.integers(0, 14, (NQ, 1, M, 16)).astype(np.int8); assign = np.zeros((NQ, 1), np.int32)
configs = []
for spec in sys.argv[1:]:
    v, w = spec.split(":"); configs.append((int(v, 0), int(w)))
res = {c: [] for c in configs}
for r in range(ROUNDS + 1):
    for c in configs:
        idx.set_option("variant", c[0]); idx.set_option("wgs_per_item", c[1]); idx.profile_reset()
        idx.scan_i8(assign, qt, 100)
        p = idx.profile()
        if r: res[c].append(p["scan_codes"] * (M // 2) / (p["scan_ms"] * 1e-3) / 1e9)
for c in configs:
    v = res[c]
    print("variant 0x%02x (U=%d nt=%d chunk=%d probe=%d) wgs=%4d : median %.0f GB/s  min %.0f max %.0f  (%.1f%% of 8TB/s)" % (
        c[0], [1, 2, 4, 2][c[0] & 3], (c[0] >> 2) & 1, (c[0] >> 3) & 1, (c[0] >> 4) & 1, c[1], np.median(v), min(v), max(v), np.median(v) / 80))
