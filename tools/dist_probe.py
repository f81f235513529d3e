"""Host-side timeline of one rank's multi-GPU step (submit / collect_candidates / merge) on a 1-GPU box:
world 1 over RCCL, shard size = what one of 8 ranks holds.  Shows where a rank's step time goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import torch, torch.distributed as dist
import pyqadc
from pyqadc import sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
N = int(float(os.environ.get("N", 125e6))); NQ = 8; M = 16; R = 100
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01)
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
q = rng.normal(size=(NQ, M, 1, 8)).astype(np.float32)
tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, 256), np.float32)
assign = np.zeros((NQ, 1), np.int32)
for _ in range(3):
    idx.submit(0, assign, tables.copy(), R); r = idx.collect_candidates(0); sharded.merge_batch(r, NQ, R, r["status"], dev)
pend = None; rows = []
t0 = time.perf_counter()
STEPS = int(os.environ.get('STEPS', 300))
for s in range(STEPS):
    a = time.perf_counter(); idx.submit(s % 2, assign, tables.copy(), R); b = time.perf_counter()
    if pend is not None:
        r = idx.collect_candidates(pend); c = time.perf_counter()
        sharded.merge_batch(r, NQ, R, r["status"], dev); d = time.perf_counter()
        rows.append(((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3))
    pend = s % 2
tot = (time.perf_counter() - t0) * 1e3
rows = np.array(rows[STEPS // 3:])
print("per step: submit %.3f ms, collect %.3f ms, merge %.3f ms; loop %.3f ms/step" % (*rows.mean(0), tot / STEPS))
dist.destroy_process_group()
