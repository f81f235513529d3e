"""Counters of the longest dispatch of a kernel from a rocprofv3 --pmc CSV.  usage: pmc_longest.py <dir> <kernel substring>"""
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d = per[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d["_grid"] = r.get("Grid_Size", "?")
d = max(per.values(), key=lambda c: c["_ns"])
for k in sorted(d): print("%-24s %s" % (k, d[k]))
if "GRBM_GUI_ACTIVE" in d:
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    print("clock GHz %.3f" % (cyc / d["_ns"]))
    for k in ("SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
        if k in d: print("%s / (256 CU x cycles) = %.3f" % (k, d[k] / (256 * cyc)))
