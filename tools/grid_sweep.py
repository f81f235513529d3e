"""One process: per-query time of pipelined flat-scan batches over (queries per batch) x (option values).
usage: N=1e9 python tools/grid_sweep.py option v1,v2,... [k=v ...]   (env NQS=4,8,16)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
N = int(float(os.environ.get("N", 1e9))); M = int(os.environ.get("M", 16))
NQS = [int(x) for x in os.environ.get("NQS", "4,8,16,32").split(",")]
BUDGET_MS = float(os.environ.get("BUDGET_MS", 150))
opt, vals = sys.argv[1], [float(v) for v in sys.argv[2].split(",")]
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
for kv in sys.argv[3:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
def block(k, assign, tables):
    pend = None
    t0 = time.perf_counter()
    for s in range(k):
        idx.submit(s % 2, assign, tables.copy(), 100)
        if pend is not None: idx.collect(pend)
        pend = s % 2
    idx.collect(pend)
    return (time.perf_counter() - t0) * 1e3 / k
print("ms per query; rows = queries per batch, columns = %s" % opt)
print("%6s " % "NQ" + " ".join("%9g" % v for v in vals))
for NQ in NQS:
    q = rng.normal(size=(NQ, M, 1, 128 // M)).astype(np.float32)
    tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, M * 16), np.float32)
    assign = np.zeros((NQ, 1), np.int32)
    row = []
    for v in vals:
        idx.set_option(opt, v)
        one = block(3, assign, tables)
        k = max(4, int(BUDGET_MS * 4 / one))
        row.append(min(block(k, assign, tables), block(k, assign, tables)) / NQ)
    print("%6d " % NQ + " ".join("%9.4f" % r for r in row), flush=True)
