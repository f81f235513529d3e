"""Turns the rocprofv3 CSVs written by tools/profile_r01.sh (gpurun_out/prof_<tag>_*) into the
committed summaries under profiles/: <tag>_kernel_stats.csv (verbatim --stats output),
<tag>_pmc_summary.md and <tag>_hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md §HBM: separate --pmc passes; FETCH_SIZE (KB) is doubled
on gfx950 for 16-B/lane coalesced streaming reads; WRITE_SIZE (KB) read as is."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)   # newest run wins
    assert f, pattern
    return f[-1]


shutil.copy(one("prof_%s_kt/*/*_kernel_stats.csv" % tag), os.path.join(P, "%s_kernel_stats.csv" % tag))
bench_line = [l for l in open(os.path.join(G, "prof_%s_kt.log" % tag)) if l.startswith("{")][-1]
bench = json.loads(bench_line)
open(os.path.join(P, "%s_bench_under_rocprof.json" % tag), "w").write(bench_line)


def counters(sub):
    rows = list(csv.DictReader(open(one("prof_%s_%s/*/*_counter_collection.csv" % (tag, sub)))))
    per = collections.defaultdict(dict)
    for r in rows:
        if "scan_i8_kernel" in r["Kernel_Name"] or "scan_i8_mq_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])]["_name"] = r["Kernel_Name"].split("(")[0]
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            per[int(r["Dispatch_Id"])]["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


def biggest(per, key):
    d = max(per.values(), key=lambda c: c["_dur_ns"])
    return d


fall, wall = counters("fetch"), counters("write")
avg_read = sum(c["FETCH_SIZE"] for c in fall.values()) * 1024 * 2 / len(fall)
avg_write = sum(c["WRITE_SIZE"] for c in wall.values()) * 1024 / len(wall)
f = biggest(fall, "FETCH_SIZE")
w = biggest(wall, "WRITE_SIZE")
l = biggest(counters("lds"), "SQ_LDS_IDX_ACTIVE")
cfg = bench["config"]
cs = cfg["M"] // 2
# the longest launch is the last bound level: all codes past the previous level boundary, every query of the batch
# bound-level boundaries of the planner (level_base = 512, level_growth = 4, at most 16 levels)
bounds, b = [0], 512
while b < cfg["codes"] and len(bounds) < 16:
    bounds.append(b)
    b *= 4
bounds.append(cfg["codes"])
codes = max(hi - lo for lo, hi in zip(bounds, bounds[1:])) * cfg["queries_per_step"]
algo = codes * cs
fetch_bytes = f["FETCH_SIZE"] * 1024 * 2
write_bytes = w["WRITE_SIZE"] * 1024
clk = l["GRBM_GUI_ACTIVE"] / 8 / (l["_dur_ns"] * 1e-9) / 1e9
mq = "mq" in l["_name"]
groups = (cfg["queries_per_step"] + 7) // 8
# LDS-array cycles the lookups of the launch need (MI355X_MICROARCH.md, LDS table)
lds_model = codes / cfg["queries_per_step"] * groups * cfg["M"] * 4 / 64 if mq else codes * cs * 2 / 64
summary = {
    "kernel": "%s, longest launch (last bound level, %d queries)" % (l["_name"], cfg["queries_per_step"]),
    "codes": cfg["codes"], "M": cfg["M"], "queries_per_step": cfg["queries_per_step"],
    "codes_in_launch": codes, "algorithmic_bytes": algo,
    "FETCH_SIZE_KB": f["FETCH_SIZE"], "hbm_read_bytes(FETCH_SIZE*1024*2)": fetch_bytes,
    "WRITE_SIZE_KB": w["WRITE_SIZE"], "hbm_write_bytes": write_bytes,
    "longest_launch_bytes": fetch_bytes + write_bytes, "traffic_over_algorithmic": (fetch_bytes + write_bytes) / algo,
    "launches_in_pmc_run": len(fall),
    "bytes_per_launch": avg_read + avg_write,   # average over ALL launches of the kernel, like roofline.achieved
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "duration_ms_under_pmc": f["_dur_ns"] / 1e6,
    "SQ_LDS_BANK_CONFLICT": l["SQ_LDS_BANK_CONFLICT"], "SQ_LDS_IDX_ACTIVE": l["SQ_LDS_IDX_ACTIVE"],
    "lds_conflict_fraction": l["SQ_LDS_BANK_CONFLICT"] / l["SQ_LDS_IDX_ACTIVE"],
    "lds_cycles_model(code reads x lookups x cycles / 64 lanes)": lds_model,
    "effective_clock_GHz(GRBM_GUI_ACTIVE/8/duration)": clk,
    # all LDS-array cycles of the launch over (256 CUs x launch cycles): how busy the lookup pipe was
    "lds_busy_fraction(SQ_LDS_IDX_ACTIVE/(256*GRBM_GUI_ACTIVE/8))": l["SQ_LDS_IDX_ACTIVE"] / (256 * l["GRBM_GUI_ACTIVE"] / 8),
    "valu_issue_fraction(SQ_INSTS_VALU/(256*GRBM_GUI_ACTIVE/8), 1 wave-instruction per CU clock = 1.0)":
        (l["SQ_INSTS_VALU"] / (256 * l["GRBM_GUI_ACTIVE"] / 8)) if "SQ_INSTS_VALU" in l else None,
    "sq_counters_longest_launch": {k: v for k, v in l.items() if not k.startswith("_")},
}
json.dump(summary, open(os.path.join(P, "%s_hbm_traffic.json" % tag), "w"), indent=1)
with open(os.path.join(P, "%s_pmc_summary.md" % tag), "w") as o:
    o.write("# %s — PMC summary of the dominant kernel (rocprofv3, MI355X)\n\n" % tag)
    o.write("Command per pass: `rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 2 --warmup 1` "
            "(FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set in three separate runs; tools/profile_r01.sh).\n\n")
    for k, v in summary.items():
        o.write("- %s: %s\n" % (k, v))
    o.write("\nKernel-trace stats (`--kernel-trace --stats`) are in %s_kernel_stats.csv; the bench line printed under the "
            "profiler is %s_bench_under_rocprof.json.\n" % (tag, tag))
print(json.dumps(summary, indent=1))
