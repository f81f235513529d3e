"""One configuration of tools/ivf_shard_sizes.py (for rocprofv3): python3 tools/ivf_shard_one.py c3|c5 whole|range|none [rank]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
os.environ.setdefault("QADC_BENCH_CPU_SECONDS", "0")   # (no CPU sample for these probes)
import bench
if os.environ.get('QADC_PROBE_R'): bench.R = int(os.environ['QADC_PROBE_R'])   # probe: how much of the step is replay latency
from ivf_shard_sizes import SHAPES, WORLD
# process history before the measured index exists (round 3: the figures depended on it): "destroyed" = an index was
# created and destroyed first, "alive" = another index stays alive
_hist = os.environ.get("QADC_PROBE_HISTORY", "fresh")
if _hist in ("destroyed", "alive"):
    import numpy as np, pyqadc
    _other = pyqadc.Index(16, 0)
    _other.add_partition_synthetic(100000, 1)
    _other.finalize(0.01)
    _other.query_scan(np.zeros((1, 1), np.int32), np.random.default_rng(0).random((1, 1, 256)).astype(np.float32), 10)
    if _hist == "destroyed":
        _other.close()
name, placement = sys.argv[1], sys.argv[2]
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shard = None if placement == "none" else dict(rank=r, world=WORLD, placement=placement, init=lambda ix: ix.dist_init_loopback(r, WORLD), merge="loopback")
o = bench.ivf_leg(0, shard=shard, **SHAPES[name])
print(json.dumps({k: o[k] for k in o if k not in ("workload", "algorithmic_GBps_rule") and not k.startswith("_")}))
