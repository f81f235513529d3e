"""gpurun_out/prof_<tag>_* (tools/profile_round.sh) -> profiles/<tag>_*  (file names below with tag r05):

  r05_headline_kernel_stats.csv     `rocprofv3 --kernel-trace --stats` kernel summary of `python3 bench.py --steps 20 --warmup 2` with
                                    the side legs off: the headline region (scan_i8_kernel<16,true,true,false>: 32 queries per launch,
                                    ONE QUERY PER PASS) and the batched region (scan_i8_mq_kernel).  The AverageNs of scan_i8_kernel
                                    is what `roofline.avg_launch_ms` of r05_headline_bench.json (the line of that same run) must match.
  r05_headline_bench.json           the JSON line of that run
  r05_headline_hbm_traffic.json     PMC passes of `bench.py --pmc-leg` (two steps of the headline's mode): HBM bytes
                                    (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024; separate passes — MI355X_MICROARCH.md "HBM" /
                                    "rocprofv3 PMC slots"), LDS / VALU / wait shares of the longest launch
  r05_kernel_stats.csv, r05_bench_under_rocprof.json   the default command (every leg) under the kernel trace
  r05_bench_plain.json, r05_shard_sizes.txt, r05_ivf_shard_sizes.txt
  r05_ivf_kernel_stats.csv / r05_ivfc5_kernel_stats.csv, r05_ivf_pmc_summary.json / r05_ivfc5_pmc_summary.json   (as round 4)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, TAG = os.environ.get("QADC_PROF_RAW", os.path.join(ROOT, "gpurun_out")), (sys.argv[1] if len(sys.argv) > 1 else "r06")
P = os.environ.get("QADC_PROFILES_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern), recursive=True), key=os.path.getmtime)
    assert f, pattern
    return f[-1]


def json_line(path, key=None):
    for l in reversed([l for l in open(path) if l.startswith("{")]):
        j = json.loads(l)
        if key is None or key in j:
            return j, l
    raise SystemExit("no JSON line in " + path)


def counters(sub, names):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(one("prof_%s_%s/**/*counter_collection.csv" % (TAG, sub)))):
        for n in names:
            if n in r["Kernel_Name"]:
                d = per[(n, int(r["Dispatch_Id"]))]
                d["_name"] = r["Kernel_Name"].split("(")[0]
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


# ---- (1) headline + batched regions under the kernel trace -------------------------------------------------------------
shutil.copy(one("prof_%s_headline_kt/**/*_kernel_stats.csv" % TAG), os.path.join(P, "%s_headline_kernel_stats.csv" % TAG))
bench, line = json_line(os.path.join(G, "prof_%s_headline_kt.log" % TAG), "metric")
open(os.path.join(P, "%s_headline_bench.json" % TAG), "w").write(line)
rows = list(csv.DictReader(open(os.path.join(P, "%s_headline_kernel_stats.csv" % TAG))))
agree = {}
for r in rows:
    if "scan_i8_kernel<" in r["Name"]:
        agree = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "rocprof_AverageNs": float(r["AverageNs"]),
                 "bench_roofline_avg_launch_ms": bench["roofline"]["avg_launch_ms"], "bench_roofline_launches": bench["roofline"]["launches"],
                 "note": "rocprof counts warm-up and timed steps, the line the timed ones only; a launch = one bound level of 32 queries, so the "
                         "averages agree when the per-level mix is the same (it is: every step launches the same seven levels)"}
        break

# ---- (2) PMC of the headline mode -----------------------------------------------------------------------------------------
leg, _ = json_line(os.path.join(G, "prof_%s_headline_FETCH_SIZE.log" % TAG), "pmc_leg")
cfg = bench["config"]
cs = cfg["M"] // 2
names = ["scan_i8_kernel<"]
f, w = counters("headline_FETCH_SIZE", names), counters("headline_WRITE_SIZE", names)
read_b = sum(c["FETCH_SIZE"] for c in f.values()) * 1024 * 2
write_b = sum(c["WRITE_SIZE"] for c in w.values()) * 1024
algo = leg["scan_codes"] * cs
lf = max(f.values(), key=lambda c: c["_ns"])
sq = max(counters("headline_sq", names).values(), key=lambda c: c["_ns"])
cyc = sq["GRBM_GUI_ACTIVE"] / 8
head = {
    "mode": "one_query_per_pass_batch32", "codes": cfg["codes"], "M": cfg["M"], "queries_per_step": cfg["queries_per_step"], "kernel": lf["_name"],
    "what": "the headline's mode (bench.py --pmc-leg: two steps of %d queries, every query walks the whole list by itself), all %d launches "
            "of the streaming kernel" % (cfg["queries_per_step"], len(f)),
    "launches_in_pmc_run": len(f), "launches_counted_by_the_library": leg["scan_launches"],
    "FETCH_SIZE_KB_total": sum(c["FETCH_SIZE"] for c in f.values()), "hbm_read_bytes(FETCH_SIZE*1024*2)": read_b,
    "WRITE_SIZE_KB_total": sum(c["WRITE_SIZE"] for c in w.values()), "hbm_write_bytes": write_b,
    "algorithmic_bytes(%d B x codes x queries of those launches)" % cs: algo,
    "traffic_over_algorithmic": (read_b + write_b) / algo,
    "bytes_per_launch": (read_b + write_b) / len(f), "algorithmic_bytes_per_launch": algo / leg["scan_launches"],
    "longest_launch": {"FETCH_SIZE_KB": lf["FETCH_SIZE"], "duration_ms_under_pmc": lf["_ns"] / 1e6,
                       "read_GBps_under_pmc": lf["FETCH_SIZE"] * 1024 * 2 / lf["_ns"]},
    "sq_counters_longest_launch": {
        "duration_ms_under_pmc": sq["_ns"] / 1e6, "effective_clock_GHz(GRBM_GUI_ACTIVE/8/duration)": cyc / sq["_ns"],
        "lds_conflict_fraction": sq["SQ_LDS_BANK_CONFLICT"] / max(sq["SQ_LDS_IDX_ACTIVE"], 1),
        "lds_busy_fraction(SQ_LDS_IDX_ACTIVE/(256*cycles))": sq["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
        "valu_issue_fraction(SQ_INSTS_VALU/(256*cycles))": sq["SQ_INSTS_VALU"] / (256 * cyc)},
    "live_roofline_of_the_kernel_trace_run": {k: bench["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches")},
    "kernel_trace_agreement": agree,
}
try:
    s2 = max(counters("headline_sq2", names).values(), key=lambda c: c["_ns"])
    head["sq_counters_longest_launch"]["waves_waiting_fraction(SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = s2["SQ_WAIT_ANY"] / max(s2["SQ_WAVE_CYCLES"], 1)
except (AssertionError, ValueError, KeyError):
    pass
json.dump(head, open(os.path.join(P, "%s_headline_hbm_traffic.json" % TAG), "w"), indent=1)
print(json.dumps(head, indent=1))

# ---- (4) default command under the kernel trace; plain line; stand-ins ----------------------------------------------------
try:
    shutil.copy(one("prof_%s_kt/**/*_kernel_stats.csv" % TAG), os.path.join(P, "%s_kernel_stats.csv" % TAG))
    open(os.path.join(P, "%s_bench_under_rocprof.json" % TAG), "w").write(json_line(os.path.join(G, "prof_%s_kt.log" % TAG), "metric")[1])
except (AssertionError, SystemExit) as e:
    print("skipped the whole-bench trace:", e)
for name in ("bench_plain.json", "shard_sizes.txt", "ivf_shard_sizes.txt"):
    src = os.path.join(G, "%s_%s" % (TAG, name))
    if os.path.exists(src):
        if name.endswith(".json"):
            lines = [l for l in open(src) if l.startswith("{")]
            if lines:
                open(os.path.join(P, "%s_%s" % (TAG, name)), "w").write(lines[-1])
        else:
            shutil.copy(src, os.path.join(P, "%s_%s" % (TAG, name)))

# ---- (5) IVF legs (round 4's summary, tag r05) ---------------------------------------------------------------------------
X = "ivf"


def shape_summary(X_, SHAPE):
    global X
    X = X_
    shutil.copy(one("prof_%s_%s_kt/**/*_kernel_stats.csv" % (TAG, X)), os.path.join(P, "%s_%s_kernel_stats.csv" % (TAG, X)))
    leg = json.loads([l for l in open(os.path.join(G, "prof_%s_%s_kt.log" % (TAG, X))) if l.startswith("{")][-1])
    names = {"scan_query_kernel": "head (front + first probes of every query, one workgroup per query)",
             "scan_i8_mq_narrow_kernel": "partition-major second phase (8 or 4 queries per pass)",
             "replay_heap_wave_kernel": "wave-per-query heap replay", "order_cands_kernel": "scan order per query"}
    out = {"command": "python3 tools/ivf_shard_one.py %s none  (bench.py's IVF leg of that shape alone; 1024- and 2048-query batches)" % SHAPE,
           "us_per_query_in_the_kernel_trace_run": leg["us_per_query"], "kernels": {}}
    f, w, sq, sq2 = (counters("%s_%s" % (X, k), names) for k in ("FETCH_SIZE", "WRITE_SIZE", "sq", "sq2"))
    for n, what in names.items():
        fd = [v for (k, _), v in f.items() if k == n]
        wd = [v for (k, _), v in w.items() if k == n]
        sd = [v for (k, _), v in sq.items() if k == n]
        s2 = [v for (k, _), v in sq2.items() if k == n]
        if not fd or not sd:
            continue
        med = sorted(d["_ns"] for d in sd)[len(sd) // 2]              # 1024-query launches only (the run also has 2048-query batches)
        pick = lambda ds: [d for d in ds if d["_ns"] <= med * 1.4] or ds
        fd, wd, sd, s2 = pick(fd), pick(wd), pick(sd), pick(s2)
        avg = lambda ds, k: sum(d.get(k, 0.0) for d in ds) / max(len(ds), 1)
        cyc = avg(sd, "GRBM_GUI_ACTIVE") / 8
        ent = {"what": what, "launches_averaged": len(sd), "avg_duration_ms_under_pmc": avg(sd, "_ns") / 1e6,
               "hbm_read_bytes_per_launch(FETCH_SIZE*1024*2)": avg(fd, "FETCH_SIZE") * 2048, "hbm_write_bytes_per_launch": avg(wd, "WRITE_SIZE") * 1024,
               "hbm_GBps_under_pmc": (avg(fd, "FETCH_SIZE") * 2048 + avg(wd, "WRITE_SIZE") * 1024) / max(avg(fd, "_ns"), 1),
               "effective_clock_GHz": cyc / max(avg(sd, "_ns"), 1),
               "lds_busy_fraction(SQ_LDS_IDX_ACTIVE/(256*cycles))": avg(sd, "SQ_LDS_IDX_ACTIVE") / (256 * cyc) if cyc else None,
               "lds_conflict_fraction": avg(sd, "SQ_LDS_BANK_CONFLICT") / max(avg(sd, "SQ_LDS_IDX_ACTIVE"), 1),
               "valu_issue_fraction(SQ_INSTS_VALU/(256*cycles))": avg(sd, "SQ_INSTS_VALU") / (256 * cyc) if cyc else None}
        if s2:
            wc = avg(s2, "SQ_WAVE_CYCLES")
            ent["waves_waiting_fraction(SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = avg(s2, "SQ_WAIT_ANY") / wc if wc else None
            ent["waves_waiting_on_lds_fraction(SQ_WAIT_INST_LDS/SQ_WAVE_CYCLES)"] = avg(s2, "SQ_WAIT_INST_LDS") / wc if wc else None
        out["kernels"][n] = ent
    json.dump(out, open(os.path.join(P, "%s_%s_pmc_summary.json" % (TAG, X)), "w"), indent=1)
    print(json.dumps(out, indent=1)[:2500])


for X_, SHAPE in (("ivf", "c3"), ("ivfc5", "c5")):
    try:
        shape_summary(X_, SHAPE)
    except (AssertionError, IndexError) as e:
        print("skipped", X_, e)
