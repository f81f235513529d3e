"""Per-kernel timeline of one steady-state batch from a rocprofv3 --kernel-trace CSV.
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [batch index from the end, default 3]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a batch starts at the first start_scan / select kernel after a sort
starts = [i for i, r in enumerate(rows) if "sort_cands" in r["Kernel_Name"]]
a, b = starts[-back - 1] + 1, starts[-back] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a - 1:b + 3]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    print("%9.1f us  dur %8.1f  gap %6.1f  grid %8s wg %5s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
          r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name))
    prev_end = max(prev_end, e)
