"""Synchronous single query (bench.py's latency leg), launch per query against the resident kernel, same process, interleaved.
Regenerates profiles/r06_latency_resident.txt:  python tools/latency_resident.py > profiles/r06_latency_resident.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import bench
import pyqadc
import ctypes as C

def run(n, M=16, R=100, reps=400, rounds=3):
    idx = pyqadc.Index(M, 0)
    idx.add_partition_synthetic(n, 1)
    idx.finalize(bench.KEEP)
    rng = np.random.default_rng(0)
    cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
    tb = bench.make_tables(rng, cb, 1)
    a = np.zeros((1, 1), np.int32)
    keys, vals = np.zeros((1, R), np.uint32), np.zeros((1, R), np.int8)
    sizes, status = np.zeros(1, np.int32), np.zeros(1, np.int32)
    qmin, qmax = np.zeros(1, np.float32), np.zeros(1, np.float32)
    P = pyqadc._p
    fixed = (P(keys, pyqadc.u32p), P(vals, pyqadc.i8p), P(sizes, pyqadc.i32p), P(status, pyqadc.i32p),
             P(qmin, pyqadc.f32p), P(qmax, pyqadc.f32p), None)
    pa = P(a, pyqadc.i32p)
    fn, h = pyqadc.lib().qadc_query_scan, idx._h
    copies = [tb.copy() for _ in range(reps)]
    ptrs = [P(t, pyqadc.f32p) for t in copies]
    out = {}
    ref = None
    for rnd in range(rounds):
        for mode in (0, 1):
            idx.set_option("resident", mode)
            ts = []
            for i in range(reps):
                copies[i][...] = tb
                t0 = time.perf_counter()
                rc = fn(h, 1, 1, pa, ptrs[i], R, *fixed)
                t1 = time.perf_counter()
                assert rc == 0 and status[0] == 0 and sizes[0] == R
                if ref is None:
                    ref = (keys.copy(), vals.copy())
                assert np.array_equal(ref[0], keys) and np.array_equal(ref[1], vals)
                if i >= 20:
                    ts.append(t1 - t0)
            ts = np.sort(np.array(ts)) * 1e6
            out.setdefault(mode, []).append((float(np.median(ts)), float(ts[len(ts) // 10]), float(ts[len(ts) * 9 // 10])))
    pr = idx.profile()
    idx.close()
    for mode in (0, 1):
        print("codes %8d  %-18s median / p10 / p90 us per round: %s" % (n, "resident kernel" if mode else "launch per query",
              "  ".join("%.1f / %.1f / %.1f" % t for t in out[mode])))
    print("   resident launches %d queries %d fallbacks %d" % (pr["resident_launches"], pr["resident_queries"], pr["resident_fallbacks"]))

if __name__ == "__main__":
    import gc
    gc.disable()
    for n in (100000, 30000, 250000):
        run(n)
