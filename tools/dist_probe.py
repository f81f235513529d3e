"""Host-side timeline of one rank's multi-GPU step on a 1-GPU box (world 1 over RCCL, shard size = what one of 8
ranks holds): where the host time of the pipelined loop of bench.py (run_steps_dist) goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
import torch, torch.distributed as dist
import pyqadc
from pyqadc import sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
N = int(float(os.environ.get("N", 125e6))); NQ = int(os.environ.get("NQ", 32)); M = 16; R = 100
STEPS = int(os.environ.get("STEPS", 300)); WORLD_EMU = int(os.environ.get("WORLD_EMU", 8))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
idx = pyqadc.Index(M); idx.add_partition_synthetic(N, 1); idx.finalize(0.01 * 1e9 / N if N < 1e9 else 0.01)
for kv in sys.argv[1:]:
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
rng = np.random.default_rng(0)
cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
q = rng.normal(size=(NQ, M, 1, 8)).astype(np.float32)
tables = np.ascontiguousarray(((q - cb[None]) ** 2).sum(-1).reshape(NQ, 1, 256), np.float32)
assign = np.zeros((NQ, 1), np.int32)
# this rank pre-scans 1/WORLD_EMU of the (10M) starts and replays NQ/WORLD_EMU queries, as one of WORLD_EMU ranks would
T = {k: 0.0 for k in ("prescan_submit", "collect_cand", "prescan_collect", "merge", "submit")}
def tick(k, t0):
    T[k] += time.perf_counter() - t0
# as one of WORLD_EMU ranks: this rank replays only every WORLD_EMU-th query of the gathered streams
_orig_merge = sharded.merge_streams_i8
sharded.merge_streams_i8 = lambda g, world, nq, R_, cap, ma, qf, qs, *rest: _orig_merge(g, world, nq, R_, cap, ma, 0, WORLD_EMU, *rest)
LEAD = int(os.environ.get("LEAD", 3))
tbs = {}
def prescan(b):
    tbs[b % 6] = tables.copy()
    idx.prescan_submit(b % 2, assign, tbs[b % 6], R, 0, WORLD_EMU)
for b in range(LEAD):
    prescan(b)
    idx.submit(b % 4, assign, tbs[b % 6], R, prescan=np.tile(idx.prescan_collect(b % 2), (1, WORLD_EMU)))
prescan(LEAD)
t_start = None; s0 = 0
for i in range(STEPS):
    if i == STEPS // 3:
        for k in T: T[k] = 0.0
        t_start = time.perf_counter(); s0 = i
    t0 = time.perf_counter(); prescan(i + LEAD + 1); tick("prescan_submit", t0)
    t0 = time.perf_counter(); res = idx.collect_candidates(i % 4); tick("collect_cand", t0)
    t0 = time.perf_counter(); pv = idx.prescan_collect((i + LEAD) % 2); tick("prescan_collect", t0)
    t0 = time.perf_counter(); out = sharded.merge_batch(res, NQ, R, res["status"], dev, extra=pv); tick("merge", t0)
    t0 = time.perf_counter(); idx.submit((i + LEAD) % 4, assign, tbs[(i + LEAD) % 6], R, prescan=np.tile(out[3], (1, WORLD_EMU))); tick("submit", t0)
tot = (time.perf_counter() - t_start) * 1e3 / (STEPS - s0)
print("per step (ms): " + ", ".join("%s %.3f" % (k, v * 1e3 / (STEPS - s0)) for k, v in T.items()) + "; loop %.3f" % tot)
for b in range(STEPS, STEPS + LEAD):
    idx.collect_candidates(b % 4)
for sl in (0, 1):
    try:
        idx.prescan_collect(sl)
    except Exception:
        pass
dist.destroy_process_group()
