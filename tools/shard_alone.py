"""Standalone durations of one 32-query batch's kernels on a 125M-code shard (nothing else in flight): run under
rocprofv3 --kernel-trace.  Modes: plain batch; sliced pre-scan (1/8) + injected batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
N, W, NQ, R, M = int(1e9), 8, 32, 100, 16
idx = pyqadc.Index(M)
idx.add_partition_synthetic_shard(N, 0, N // W // 16 * 16, 0x5EED0001, max(1, int(np.float32(N) * np.float32(0.01))))
idx.finalize(0.01)
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
q = rng.normal(size=(NQ, M, 1, 8)).astype(np.float32)
tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, 256), np.float32)
assign = np.zeros((NQ, 1), np.int32)
for rep in range(3):
    idx.query_scan(assign, tables.copy(), R)
for rep in range(3):
    idx.prescan_submit(0, assign, tables.copy(), R, 0, W)
    pv = idx.prescan_collect(0)
    idx.submit(0, assign, tables.copy(), R, prescan=np.tile(pv, (1, W)))
    idx.collect(0)
idx.close()
