"""IVF at the C3 shape, synchronous batches of a few dozen to a few hundred queries: which path is faster?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M, K, MA, dim, N, R = 16, 4096, 32, 128, 100_000_000, 100
rng = np.random.default_rng(0)
sizes = rng.multinomial(N, np.ones(K) / K)
idx = pyqadc.Index(M)
for p in range(K):
    idx.add_partition_synthetic(int(sizes[p]), 1000 + p)
idx.finalize(0.01)
idx.set_pq(rng.normal(size=(M, 16, dim // M)).astype(np.float32))
idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
for nq in [int(x) for x in os.environ.get('NQS', '1,4,16,32,64,128,192,256').split(',')]:
    q = rng.normal(size=(nq, dim)).astype(np.float32)
    out, p90 = [], []
    for opts in ([{"device_replay_alone_nq": 400}, {"device_replay_alone_nq": 100000}, {"device_replay_alone_nq": 0}] if os.environ.get("REPLAY_AB") else [{"wgq": 1}, {"wgq": 0}, {"wgq": 2}]):
        for k, v in opts.items():
            idx.set_option(k, v)
        for _ in range(3):
            idx.search(q, MA, R)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            idx.search(q, MA, R)
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts)) * 1e3
        out.append(float(np.median(ts)))
        p90.append(float(ts[36]))
    print("nq %4d: auto %.3f ms  level path %.3f ms  query kernel %.3f ms   (p90 %.3f / %.3f / %.3f)" % (
        nq, out[0], out[1], out[2], p90[0], p90[1], p90[2]))
