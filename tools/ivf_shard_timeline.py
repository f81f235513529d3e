"""Per-queue kernel timeline of a steady-state window of tools/ivf_shard_trace.sh's trace.
usage: python tools/ivf_shard_timeline.py <dir> [window_ms=4.5] [anchor_index=36] [anchor_kernel=replay_heap_wave]
(bench.ivf_leg: 4 sizing + 8 warm + 48 timed 1024-query batches, then the 2048-query and the profiled passes)"""
import csv, glob, os, sys
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 4.5
first = int(sys.argv[3]) if len(sys.argv) > 3 else 36
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region = the densest run of replay kernels; take the window ending `back` ms before the last replay
anchor = sys.argv[4] if len(sys.argv) > 4 else "replay_heap_wave"
rep = [r for r in rows if anchor in r["Kernel_Name"]]
t_beg = int(rep[first]["Start_Timestamp"])
t_end = t_beg + int(win * 1e6)
queues = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t_beg or s > t_end:
        continue
    q = r.get("Queue_Id", "?")
    queues.setdefault(q, len(queues))
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("qadc::", "").split("(")[0][:44]
    print("%9.1f us  dur %7.1f  q%-2d grid %8s  %s" % ((s - t_beg) / 1e3, (e - s) / 1e3, queues[q], r.get("Grid_Size_X", r.get("Grid_Size", "?")), name))
busy = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t_beg or e > t_end:
        continue
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("qadc::", "").split("(")[0][:44]
    busy[k] = busy.get(k, 0) + (e - s) / 1e3
print("--- busy us in the %.1f ms window, by kernel" % win)
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("%9.1f  %s" % (v, k))
