"""IVF-shaped probe: K synthetic partitions, nprobe partitions per query, batched queries."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import pyqadc
M = int(os.environ.get("M", 16)); N = int(float(os.environ.get("N", 1e8))); K = int(os.environ.get("K", 4096))
MA = int(os.environ.get("MA", 32)); NQ = int(os.environ.get("NQ", 256)); R = 100
rng = np.random.default_rng(0)
sizes = rng.multinomial(N, np.ones(K) / K)
idx = pyqadc.Index(M)
t = time.time()
for p in range(K):
    idx.add_partition_synthetic(int(sizes[p]), 1000 + p)
idx.finalize(0.01); idx.set_option("profile", 1)
for opt in sys.argv[1:]:
    k, v = opt.split("="); idx.set_option(k, float(v))
print("built %d partitions (%d codes) in %.1fs" % (K, N, time.time() - t), flush=True)
cb = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
def tables(nq):
    q = rng.normal(size=(nq * MA, M, 1, 128 // M)).astype(np.float32)
    return np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(nq, MA, M * 16), np.float32)
assign = np.stack([rng.choice(K, MA, replace=False) for _ in range(NQ)]).astype(np.int32)
tb = tables(NQ)
idx.query_scan(assign, tb.copy(), R)
for r in range(3):
    idx.profile_reset(); t = time.time()
    res = idx.query_scan(assign, tb.copy(), R)
    dt = time.time() - t; p = idx.profile()
    codes = sizes[assign].sum()
    print("rep %d: %.2f ms/batch (%.1f us/query) -> %.3e codes/s | scan kernels %.2f ms (%d launches, %.0f GB/s) small %.2f ms (%d, %.0f GB/s) prescan %.2f ms "
          "host asm %.2f plan %.2f heap %.2f ms cands/query %.0f max %d hostsorted %d regrows %d status_bad %d" % (r, dt * 1e3, dt * 1e6 / NQ, codes / dt, p["scan_ms"], p["scan_launches"],
          p["scan_codes"] * (M // 2) / (max(p["scan_ms"], 1e-9) * 1e-3) / 1e9, p["small_ms"], p["small_launches"],
          p["small_codes"] * (M // 2) / (max(p["small_ms"], 1e-9) * 1e-3) / 1e9, p["start_ms"], p["host_replay_ms"], p["host_plan_ms"], p["host_heap_ms"], p["candidates"] / NQ, 0, p["host_sorted_queries"], p["regrows"],
          int(res["status"].sum())), flush=True)

# ---- N1 path: queries in (coarse assignment + tables on the GPU) ----
dim = 128
coarse = rng.normal(size=(K, dim)).astype(np.float32)
idx.set_pq(cb); idx.set_coarse(coarse)
queries = rng.normal(size=(NQ, dim)).astype(np.float32)
idx.search(queries, MA, R)
for r in range(3):
    idx.profile_reset(); t = time.time()
    res = idx.search(queries, MA, R)
    dt = time.time() - t; p = idx.profile()
    codes = sizes[res["assign"]].sum()
    print("search rep %d: %.2f ms/batch (%.1f us/query) -> %.3e codes/s | prescan+tables %.2f ms host asm %.2f plan %.2f heap %.2f ms "
          "cands/query %.0f" % (r, dt * 1e3, dt * 1e6 / NQ, codes / dt, p["start_ms"], p["host_replay_ms"], p["host_plan_ms"],
                                p["host_heap_ms"], p["candidates"] / NQ), flush=True)

# ---- the same, pipelined through the two slots (batch s+1 is enqueued before batch s is collected) ----
for nqb in (256, 1024):
    qs = [rng.normal(size=(nqb, dim)).astype(np.float32) for _ in range(4)]
    idx.search_submit(0, qs[0], MA, R); idx.search_collect(0)
    steps = 12
    idx.profile_reset(); t = time.time(); pend = None; ncodes = 0
    for s_ in range(steps):
        idx.search_submit(s_ % 2, qs[s_ % 4], MA, R)
        if pend is not None:
            ncodes += sizes[idx.search_collect(pend)["assign"]].sum()
        pend = s_ % 2
    ncodes += sizes[idx.search_collect(pend)["assign"]].sum()
    dt = time.time() - t; p = idx.profile()
    print("search pipelined, %d queries per batch: %.2f ms/batch (%.2f us/query) -> %.3e codes/s | per batch: plan %.2f asm %.2f heap %.2f ms"
          % (nqb, dt * 1e3 / steps, dt * 1e6 / steps / nqb, ncodes / dt, p["host_plan_ms"] / steps, p["host_replay_ms"] / steps,
             p["host_heap_ms"] / steps), flush=True)
