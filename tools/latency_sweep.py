"""Synchronous single-query latency (bench.py's latency_leg) under option sets: python tools/latency_sweep.py "opt=v,opt=v" ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc, bench
R, KEEP = 100, 0.01
for opts in (sys.argv[1:] or [""]):
    M, n = 16, int(os.environ.get("LAT_CODES", 100000))
    idx = pyqadc.Index(M, 0); idx.add_partition_synthetic(n, 1); idx.finalize(KEEP)
    for kv in filter(None, opts.split(",")):
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    rng = np.random.default_rng(0)
    cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
    tb = bench.make_tables(rng, cb, 1); a = np.zeros((1, 1), np.int32)
    for _ in range(20): idx.query_scan(a, tb.copy(), R)
    keys, vals = np.zeros((1, R), np.uint32), np.zeros((1, R), np.int8)
    sizes, status = np.zeros(1, np.int32), np.zeros(1, np.int32)
    qmin, qmax = np.zeros(1, np.float32), np.zeros(1, np.float32)
    P = pyqadc._p
    fixed = (P(keys, pyqadc.u32p), P(vals, pyqadc.i8p), P(sizes, pyqadc.i32p), P(status, pyqadc.i32p), P(qmin, pyqadc.f32p), P(qmax, pyqadc.f32p), None)
    pa = P(a, pyqadc.i32p); fn, h = pyqadc.lib().qadc_query_scan, idx._h
    copies = [tb.copy() for _ in range(320)]; ptrs = [P(t, pyqadc.f32p) for t in copies]
    ts = []
    for i in range(320):
        t0 = time.perf_counter(); rc = fn(h, 1, 1, pa, ptrs[i], R, *fixed); t1 = time.perf_counter()
        assert rc == 0 and status[0] == 0
        if i >= 20: ts.append(t1 - t0)
    ts = np.sort(np.array(ts)) * 1e6
    print("[%s] median %.1f us  p10 %.1f  p90 %.1f" % (opts, np.median(ts), ts[len(ts) // 10], ts[len(ts) * 9 // 10]), flush=True)
    idx.close()
