"""Where a head workgroup's cycles go (the query kernel's own counters): python3 tools/ivf_head_cycles.py c3|c5 [rank8] [opt=value ...]
rank8 = rank 0 of 8 under the range placement with the loopback merge (bench.ivf_leg's shard form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench, pyqadc
from ivf_shard_sizes import SHAPES
kw = SHAPES[sys.argv[1]]
M, K, MA, dim, N = kw["M"], kw["K"], kw["MA"], kw["dim"], kw["N"]
rng = np.random.default_rng(11)
sizes = rng.multinomial(N, np.ones(K) / K).astype(np.int64)
idx = pyqadc.Index(M, 0)
RANK8 = "rank8" in sys.argv
if RANK8:
    sys.argv.remove("rank8")
for p in range(K):
    n = int(sizes[p])
    if RANK8 and n:
        per = (n // 8) // 16 * 16
        idx.add_partition_synthetic_shard(n, 0, per, kw["seed0"] + p, max(1, int(np.float32(n) * np.float32(bench.KEEP))))
    else:
        idx.add_partition_synthetic(n, kw["seed0"] + p)
idx.finalize(bench.KEEP)
idx.set_pq(rng.normal(size=(M, 16, dim // M)).astype(np.float32))
idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
if RANK8:
    idx.dist_init_loopback(0, 8)
for kv in sys.argv[2:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
q = rng.normal(size=(1024, dim)).astype(np.float32)
def one():
    if RANK8:
        idx.search_submit(0, q, MA, bench.R)
        idx.dist_collect(0)
    else:
        idx.search(q, MA, bench.R)
for _ in range(3):
    one()
idx.set_option("profile", 1)
idx.profile_reset()
for _ in range(8):
    one()
p = idx.profile()
nqs = max(p["wgq_queries"], 1)
print("%s: per query (cycles of the head workgroup): front %.0f  scan %.0f  sort %.0f | per batch: head %.3f ms  grouped %.3f ms  order %.3f ms" % (
    sys.argv[1], p["wgq_front_cycles"] / nqs, p["wgq_scan_cycles"] / nqs, p["wgq_sort_cycles"] / nqs,
    p["group_head_ms"] / max(p["group_batches"], 1), p["group_scan_ms"] / max(p["group_batches"], 1), p["group_order_ms"] / max(p["group_batches"], 1)))
