"""Which hardware queue every HIP stream of a run landed on, from a rocprofv3 kernel trace:
    python3 tools/queue_map.py <trace dir>   ->  one line per (Stream_Id, Queue_Id): dispatches, the kernels seen there"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
for f in files:
    per = defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("qadc::", "").split("(")[0].split("<")[0]
        per[(int(row["Stream_Id"]), int(row["Queue_Id"]))][name] += 1
    if not per:
        continue
    print(os.path.basename(f))
    byq = defaultdict(list)
    for (s, q), k in sorted(per.items()):
        byq[q].append(s)
        top = sorted(k.items(), key=lambda kv: -kv[1])[:6]
        print("  stream %3d -> queue %3d  %6d dispatches: %s" % (s, q, sum(k.values()), ", ".join("%s x%d" % kv for kv in top)))
    shared = {q: s for q, s in byq.items() if len(s) > 1}
    print("  queues shared by several streams:", shared if shared else "none")
