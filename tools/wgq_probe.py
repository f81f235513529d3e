"""A/B of the one-workgroup-per-query kernel's tuning variants on the IVF shape (BASELINE configs[2]) and on the
single-query latency point, inside ONE process.  usage: python tools/wgq_probe.py [variant ...] [k=v ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = int(os.environ.get("M", 16)); N = int(float(os.environ.get("N", 1e8))); K = int(os.environ.get("K", 4096))
MA = int(os.environ.get("MA", 32)); NQB = int(os.environ.get("NQ", 1024)); R = 100; dim = int(os.environ.get("DIM", 128))
variants = [int(a) for a in sys.argv[1:] if "=" not in a] or [0, 1, 2]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
rng = np.random.default_rng(0)
sizes = rng.multinomial(N, np.ones(K) / K)
idx = pyqadc.Index(M)
for p in range(K):
    idx.add_partition_synthetic(int(sizes[p]), 1000 + p)
idx.finalize(0.01); idx.set_option("profile", 1)
for k, v in opts: idx.set_option(k, float(v))
cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
idx.set_pq(cb); idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
qs = [rng.normal(size=(NQB, dim)).astype(np.float32) for _ in range(4)]
def run(steps, depth=3):
    pend, nc = [], 0
    t0 = time.perf_counter()
    for s in range(steps):
        idx.search_submit(s % depth, qs[s % 4], MA, R); pend.append(s % depth)
        if len(pend) == depth: nc += int(sizes[idx.search_collect(pend.pop(0))["assign"]].sum())
    while pend: nc += int(sizes[idx.search_collect(pend.pop(0))["assign"]].sum())
    return time.perf_counter() - t0, nc
for rep in range(2):
    for v in variants + [-1]:
        if v < 0: idx.set_option("wgq", 0)
        else: idx.set_option("wgq", 1); idx.set_option("wgq_variant", v)
        run(3); idx.profile_reset()
        steps = 16
        dt, nc = run(steps); p = idx.profile()
        nqq = max(p["wgq_queries"], 1)
        print("ivf %s: %.3f us/query  %.2f TB/s | kernel %.3f ms/batch (%d launches) plan %.3f heap %.3f ms/batch | per query workgroup: front %.0f kcyc scan %.0f kcyc sort %.0f kcyc" % (
            "levels   " if v < 0 else "variant %d" % v, dt * 1e6 / (steps * NQB), nc * (M // 2) / dt / 1e12,
            p["wgq_ms"] / max(p["wgq_launches"], 1), p["wgq_launches"], p["host_plan_ms"] / steps, p["host_heap_ms"] / steps,
            p["wgq_front_cycles"] / nqq / 1e3, p["wgq_scan_cycles"] / nqq / 1e3, p["wgq_sort_cycles"] / nqq / 1e3), flush=True)
idx.close()
# ---- latency point ----
idx = pyqadc.Index(16); idx.add_partition_synthetic(100000, 1); idx.finalize(0.01); idx.set_option("profile", 1)
for k, v in opts: idx.set_option(k, float(v))
cb = rng.normal(size=(16, 16, 8)).astype(np.float32)
q = rng.normal(size=(1, 16, 1, 8)).astype(np.float32)
tb = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(1, 1, 256), np.float32)
a = np.zeros((1, 1), np.int32)
for v in variants + [-1]:
    if v < 0: idx.set_option("wgq", 0)
    else: idx.set_option("wgq", 1); idx.set_option("wgq_variant", v)
    for _ in range(20): idx.query_scan(a, tb.copy(), R)
    idx.profile_reset()
    ts = []
    for _ in range(200):
        t = tb.copy(); t0 = time.perf_counter(); idx.query_scan(a, t, R); ts.append(time.perf_counter() - t0)
    p = idx.profile()
    print("latency %s: median %.1f us p10 %.1f | kernel %.1f us  cands %.0f | front %.1f kcyc scan %.1f kcyc sort %.1f" % ("levels   " if v < 0 else "variant %d" % v,
          np.median(ts) * 1e6, np.sort(ts)[20] * 1e6, p["wgq_ms"] * 1e3 / max(p["wgq_launches"], 1), p["candidates"] / 200,
          p["wgq_front_cycles"] / max(p["wgq_queries"], 1) / 1e3, p["wgq_scan_cycles"] / max(p["wgq_queries"], 1) / 1e3,
          p["wgq_sort_cycles"] / max(p["wgq_queries"], 1) / 1e3), flush=True)
