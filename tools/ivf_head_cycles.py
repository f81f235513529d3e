"""Where a head workgroup's cycles go (the query kernel's own counters): python3 tools/ivf_head_cycles.py c3|c5 [opt=value ...]"""
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
for p in range(K):
    idx.add_partition_synthetic(int(sizes[p]), kw["seed0"] + p)
idx.finalize(bench.KEEP)
idx.set_pq(rng.normal(size=(M, 16, dim // M)).astype(np.float32))
idx.set_coarse(rng.normal(size=(K, dim)).astype(np.float32))
for kv in sys.argv[2:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
q = rng.normal(size=(1024, dim)).astype(np.float32)
for _ in range(3):
    idx.search(q, MA, bench.R)
idx.set_option("profile", 1)
idx.profile_reset()
for _ in range(8):
    idx.search(q, MA, bench.R)
p = idx.profile()
nqs = max(p["wgq_queries"], 1)
print("%s: per query (cycles of the head workgroup): front %.0f  scan %.0f  sort %.0f | per batch: head %.3f ms  grouped %.3f ms  order %.3f ms" % (
    sys.argv[1], p["wgq_front_cycles"] / nqs, p["wgq_scan_cycles"] / nqs, p["wgq_sort_cycles"] / nqs,
    p["group_head_ms"] / max(p["group_batches"], 1), p["group_scan_ms"] / max(p["group_batches"], 1), p["group_order_ms"] / max(p["group_batches"], 1)))
