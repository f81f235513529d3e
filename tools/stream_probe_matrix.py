"""Round 5: qadc_stream_probe over every ordered pair of the library's seven streams — microseconds a one-wave marker on stream b
waits behind a CU-hungry launch on stream a — in two process histories:  python3 tools/stream_probe_matrix.py [torch_before]
(torch_before: a torch.distributed "nccl" group with one all-reduce exists BEFORE the stream set is created)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "torch_before":
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
import pyqadc
names = ["scan", "copy", "order", "front", "alt", "coll", "merge"]
pyqadc.device_prepare(0)
pyqadc.stream_probe(0, 1)                                   # warm-up (module load, LDS opt-in)
print("history: %s   rows = stream a (the long launch), columns = stream b (the marker); marker wait in us (launch duration %s)" % (mode, "~120 us"))
print("        " + " ".join("%7s" % n for n in names))
for a in range(7):
    row = []
    for b in range(7):
        if a == b:
            row.append("      -")
            continue
        w = min(pyqadc.stream_probe(a, b)[0] for _ in range(3))
        row.append("%7.1f" % w)
    print("%7s " % names[a] + " ".join(row))
