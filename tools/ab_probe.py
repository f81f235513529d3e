"""A/B of one option inside ONE process (same box, same clocks): alternates the option every BLOCK pipelined steps.
usage: N=125e6 python tools/ab_probe.py option valueA valueB [more k=v ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
N = int(float(os.environ.get("N", 125e6))); NQ = int(os.environ.get("NQ", 8)); M = int(os.environ.get("M", 16))
BLOCK = int(os.environ.get("BLOCK", 100)); ROUNDS = int(os.environ.get("ROUNDS", 5))
opt, va, vb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
for kv in sys.argv[4:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
q = rng.normal(size=(NQ, M, 1, 128 // M)).astype(np.float32)
tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, M * 16), np.float32)
assign = np.zeros((NQ, 1), np.int32)
DEPTH = int(os.environ.get("DEPTH", 3))      # batches in flight (slots used), as bench.py
def block(k):
    pend = []
    t0 = time.perf_counter()
    for s in range(k):
        idx.submit(s % DEPTH, assign, tables.copy(), 100)
        pend.append(s % DEPTH)
        if len(pend) == DEPTH: idx.collect(pend.pop(0))
    while pend: idx.collect(pend.pop(0))
    return (time.perf_counter() - t0) * 1e3 / k
block(50)
res = {va: [], vb: []}
for r in range(ROUNDS):
    for v in (va, vb):
        idx.set_option(opt, v); block(5); res[v].append(block(BLOCK))
for v in (va, vb):
    print("%s=%g: %s -> median %.4f ms/step" % (opt, v, " ".join("%.3f" % x for x in res[v]), float(np.median(res[v]))))
