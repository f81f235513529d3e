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
# a LIVE RCCL communicator in the process (round 5): does a communicator's own queues land on the scan's or the
# front's compute pipe?  "torch_before" = torch.distributed's nccl group (torch's bundled RCCL), one all_reduce, BEFORE the library's
# stream set exists; "lib_after" = the library's own dlopen-ed RCCL (qadc_dist_init, world 1) on a holder index, i.e. right AFTER
# the stream set was created by that index; "lib_after_torch_before" = both.  The measured index then merges through the loopback
# stand-in as always (a world-1 communicator cannot carry an 8-rank merge).
_rccl = os.environ.get("QADC_PROBE_RCCL", "none")
if "torch_cuda_only" in _rccl:                               # torch's HIP runtime up (null stream only), no communicator
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
if "set_first" in _rccl:                                     # the library's stream set exists BEFORE torch creates its communicator
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    import pyqadc
    pyqadc.device_prepare(0)                                 # (what bench.py does before dist.init_process_group)
    _rccl += "_torch_before"
if "torch_before" in _rccl:
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    _t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(_t); dist.all_gather_into_tensor(torch.empty(1 << 20, device="cuda"), _t); torch.cuda.synchronize()
if "lib_after" in _rccl:
    import numpy as np, pyqadc
    _holder = pyqadc.Index(16, 0)
    _holder.add_partition_synthetic(100000, 1)
    _holder.finalize(0.01)
    _holder.dist_init(0, 1, pyqadc.dist_unique_id())
    _holder.submit(0, np.zeros((4, 1), np.int32), np.random.default_rng(0).random((4, 1, 256)).astype(np.float32), 10)
    _holder.dist_collect(0)                                  # one live all-gather through the communicator
name, placement = sys.argv[1], sys.argv[2]
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shard = None if placement == "none" else dict(rank=r, world=WORLD, placement=placement, init=lambda ix: ix.dist_init_loopback(r, WORLD), merge="loopback")
o = bench.ivf_leg(0, shard=shard, **SHAPES[name])
import pyqadc as _pq
o["stream_layout"] = _pq.stream_layout(0)
print(json.dumps({k: o[k] for k in o if k not in ("workload", "algorithmic_GBps_rule") and not k.startswith("_")}))
