"""Synchronous single query (bench.py's latency leg) with the host side of the call split by the library's own timers.
python tools/lone_latency.py [codes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import gc
gc.disable()
import numpy as np
import bench, pyqadc
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
opts = dict(kv.split("=") for kv in sys.argv[2:])                # e.g. wgq_split=24 wgq_split_codes=4096; K=64 ma=32: an IVF shape
if "lib" in opts:                                                # lib=r06base: a library built from another commit, next to the shipped one
    pyqadc.LIB_PATH = os.path.join(ROOT, "quick-adc_amd", "libqadc_hip_%s.so" % opts.pop("lib"))
K, ma = int(opts.pop("K", 1)), int(opts.pop("ma", 1))
nq = int(opts.pop("nq", 1))                                     # a small synchronous batch instead of the lone query
reps = int(opts.pop("reps", 420))                                # (lib=stamps reps=23: the phase stamps of three timed calls)            # n codes in K equal partitions, ma of them probed
M, R = 16, 100
idx = pyqadc.Index(M, 0)
for p_ in range(K):
    idx.add_partition_synthetic(n // K, 1 + p_)
idx.finalize(bench.KEEP)
for k_, v_ in opts.items():
    idx.set_option(k_, float(v_))
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
tb = bench.make_tables(rng, cb, ma * nq)
a = np.stack([rng.permutation(K)[:ma] for _ in range(nq)]).astype(np.int32).reshape(nq, ma)
keys, vals = np.zeros((nq, R), np.uint32), np.zeros((nq, R), np.int8)
sizes, status = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
qmin, qmax = np.zeros(nq, np.float32), np.zeros(nq, np.float32)
P = pyqadc._p
fixed = (P(keys, pyqadc.u32p), P(vals, pyqadc.i8p), P(sizes, pyqadc.i32p), P(status, pyqadc.i32p),
         P(qmin, pyqadc.f32p), P(qmax, pyqadc.f32p), None)
pa = P(a, pyqadc.i32p)
fn, h = pyqadc.lib().qadc_query_scan, idx._h
copies = [tb.copy() for _ in range(reps)]
ptrs = [P(t, pyqadc.f32p) for t in copies]
ts = []
for i in range(reps):
    if i == 20:
        idx.profile_reset()
    t0 = time.perf_counter()
    rc = fn(h, nq, ma, pa, ptrs[i], R, *fixed)
    t1 = time.perf_counter()
    assert rc == 0 and status[0] == 0 and sizes[0] == R
    if i >= 20:
        ts.append(t1 - t0)
pr = idx.profile()
idx.close()
ts = np.sort(np.array(ts)) * 1e6
k = reps - 20
print(opts, "K=%d ma=%d nq=%d" % (K, ma, nq), "codes %d: median %.1f us  p10 %.1f  p90 %.1f;  host timers per call: submit (plan + launch) %.2f us, stream assembly %.2f us, "
      "heap replay %.2f us; stream entries per query %.0f" % (n, np.median(ts), ts[len(ts) // 10], ts[len(ts) * 9 // 10],
      pr["host_plan_ms"] * 1e3 / k, pr["host_replay_ms"] * 1e3 / k, pr["host_heap_ms"] * 1e3 / k, pr["candidates"] / k))
