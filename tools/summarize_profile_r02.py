"""Turns the rocprofv3 CSVs written by tools/profile_r02.sh (gpurun_out/prof_<tag>_*) into the committed summaries
under profiles/ — one set per scan mode (SURVEY.md 8d keeps them apart):

  <tag>_kernel_stats.csv             verbatim `--kernel-trace --stats` kernel summary of `python3 bench.py --steps 5 --warmup 1`
  <tag>_single_kernel_stats.csv      its rows of the one-query-per-pass kernel  scan_i8_kernel<M,2,nt,chunk>
  <tag>_batched_kernel_stats.csv     its rows of the 8-queries-per-pass kernel  scan_i8_mq_kernel<M,2>
  <tag>_bench_under_rocprof.json     the JSON line bench.py printed in that profiled run
  <tag>_single_hbm_traffic.json      PMC passes of `bench.py --pmc-leg` (one query per pass)
  <tag>_batched_hbm_traffic.json     PMC passes of the headline loop (32 queries per step)
  <tag>_pmc_summary.md

HBM bytes follow MI355X_MICROARCH.md (HBM): separate --pmc passes; FETCH_SIZE (KB) is doubled on gfx950 for
16-B/lane coalesced streaming reads; WRITE_SIZE (KB) read as is."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.environ.get("QADC_PROFILES_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)   # newest run wins
    assert f, pattern
    return f[-1]


def json_line(path, key=None):
    lines = [l for l in open(path) if l.startswith("{")]
    for l in reversed(lines):
        j = json.loads(l)
        if key is None or key in j:
            return j, l
    raise SystemExit("no JSON line in " + path)


stats = one("prof_%s_kt/*/*_kernel_stats.csv" % tag)
shutil.copy(stats, os.path.join(P, "%s_kernel_stats.csv" % tag))
rows = open(stats).read().splitlines()
for mode, pat in (("single", "scan_i8_kernel<"), ("batched", "scan_i8_mq_kernel<")):
    with open(os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, mode)), "w") as o:
        o.write(rows[0] + "\n")
        for r in rows[1:]:
            if pat in r:
                o.write(r + "\n")
bench, bench_line = json_line(os.path.join(G, "prof_%s_kt.log" % tag), "metric")
open(os.path.join(P, "%s_bench_under_rocprof.json" % tag), "w").write(bench_line)


def counters(sub, names):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(one("prof_%s_%s/*/*_counter_collection.csv" % (tag, sub)))):
        if any(n in r["Kernel_Name"] for n in names):
            d = per[int(r["Dispatch_Id"])]
            d["_name"] = r["Kernel_Name"].split("(")[0]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


def sq_block(l):
    cyc = l["GRBM_GUI_ACTIVE"] / 8
    return {
        "kernel": l["_name"], "duration_ms_under_pmc": l["_dur_ns"] / 1e6,
        "effective_clock_GHz(GRBM_GUI_ACTIVE/8/duration)": cyc / l["_dur_ns"],
        "SQ_LDS_IDX_ACTIVE": l["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": l["SQ_LDS_BANK_CONFLICT"],
        "lds_conflict_fraction": l["SQ_LDS_BANK_CONFLICT"] / max(l["SQ_LDS_IDX_ACTIVE"], 1),
        "lds_busy_fraction(SQ_LDS_IDX_ACTIVE/(256*cycles))": l["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
        "valu_issue_fraction(SQ_INSTS_VALU/(256*cycles), 1 wave-instruction per CU clock = 1.0)": l["SQ_INSTS_VALU"] / (256 * cyc),
    }


out_md = ["# %s — PMC summaries of the two scan modes (rocprofv3, MI355X)\n" % tag,
          "Commands: tools/profile_r02.sh (every counter set in its own run; FETCH_SIZE and WRITE_SIZE do not fit one pass).\n"]

# ---- one query per pass -------------------------------------------------------------------------------------------
leg, _ = json_line(os.path.join(G, "prof_%s_single_FETCH_SIZE.log" % tag), "pmc_leg")
cfg = bench["config"]
cs = cfg["M"] // 2
names = ["scan_i8_kernel<"]
f, w = counters("single_FETCH_SIZE", names), counters("single_WRITE_SIZE", names)
read_b = sum(c["FETCH_SIZE"] for c in f.values()) * 1024 * 2
write_b = sum(c["WRITE_SIZE"] for c in w.values()) * 1024
algo = leg["scan_codes"] * cs
lf = max(f.values(), key=lambda c: c["_dur_ns"])
single = {
    "mode": "single", "codes": cfg["codes"], "M": cfg["M"], "kernel": lf["_name"],
    "what": "one query per pass (bench.py --pmc-leg: 3 sequential single-query batches over the whole list), "
            "all %d launches of the streaming kernel" % len(f),
    "launches_in_pmc_run": len(f), "launches_counted_by_the_library": leg["scan_launches"],
    "FETCH_SIZE_KB_total": sum(c["FETCH_SIZE"] for c in f.values()), "hbm_read_bytes(FETCH_SIZE*1024*2)": read_b,
    "WRITE_SIZE_KB_total": sum(c["WRITE_SIZE"] for c in w.values()), "hbm_write_bytes": write_b,
    "algorithmic_bytes(8 B x codes x queries of those launches)": algo,
    "traffic_over_algorithmic": (read_b + write_b) / algo,
    "bytes_per_launch": (read_b + write_b) / len(f), "algorithmic_bytes_per_launch": algo / leg["scan_launches"],
    "longest_launch": {"FETCH_SIZE_KB": lf["FETCH_SIZE"], "duration_ms_under_pmc": lf["_dur_ns"] / 1e6,
                       "read_GBps_under_pmc": lf["FETCH_SIZE"] * 1024 * 2 / lf["_dur_ns"]},
    "sq_counters_longest_launch": sq_block(max(counters("single_sq", names).values(), key=lambda c: c["_dur_ns"])),
    "live_roofline_of_the_same_head": {k: bench["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches")},
}
json.dump(single, open(os.path.join(P, "%s_single_hbm_traffic.json" % tag), "w"), indent=1)
out_md.append("\n## one query per pass — `scan_i8_kernel<%d,2,nt,chunk>`\n" % cfg["M"])
out_md += ["- %s: %s\n" % kv for kv in single.items()]

# ---- batched ------------------------------------------------------------------------------------------------------
names = ["scan_i8_mq_kernel<"]
f, w = counters("batched_FETCH_SIZE", names), counters("batched_WRITE_SIZE", names)
lf, lw = max(f.values(), key=lambda c: c["_dur_ns"]), max(w.values(), key=lambda c: c["_dur_ns"])
l = max(counters("batched_sq", names).values(), key=lambda c: c["_dur_ns"])
bounds, b = [0], 512
while b < cfg["codes"] and len(bounds) < 16:
    bounds.append(b)
    b *= 4
bounds.append(cfg["codes"])
nq = cfg["queries_per_step"]
codes = max(hi - lo for lo, hi in zip(bounds, bounds[1:])) * nq
algo = codes * cs
fb, wb = lf["FETCH_SIZE"] * 1024 * 2, lw["WRITE_SIZE"] * 1024
groups = (nq + 7) // 8
batched = {
    "mode": "batched", "codes": cfg["codes"], "M": cfg["M"], "queries_per_step": nq, "kernel": lf["_name"],
    "what": "the headline loop (32 queries per step, 8 per pass, passes as L2-sharing siblings), longest launch = last bound level",
    "codes_x_queries_in_launch": codes, "algorithmic_bytes": algo,
    "FETCH_SIZE_KB": lf["FETCH_SIZE"], "hbm_read_bytes(FETCH_SIZE*1024*2)": fb, "WRITE_SIZE_KB": lw["WRITE_SIZE"],
    "hbm_write_bytes": wb, "traffic_over_algorithmic": (fb + wb) / algo,
    "hbm_GBps_under_pmc": (fb + wb) / lf["_dur_ns"],
    "lds_cycles_model(code reads x M lookups x 4 cycles / 64 lanes)": codes / nq * groups * cfg["M"] * 4 / 64,
    "sq_counters_longest_launch": sq_block(l),
    "live_roofline_batched_of_the_same_head": {k: bench["roofline_batched"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches")},
}
json.dump(batched, open(os.path.join(P, "%s_batched_hbm_traffic.json" % tag), "w"), indent=1)
out_md.append("\n## batched — `scan_i8_mq_kernel<%d,2>`\n" % cfg["M"])
out_md += ["- %s: %s\n" % kv for kv in batched.items()]
open(os.path.join(P, "%s_pmc_summary.md" % tag), "w").writelines(out_md)
print(json.dumps({"single": single, "batched": batched}, indent=1))
