"""gpurun_out/prof_r04_* (tools/profile_r04.sh) -> profiles/r04_*  (round 3's summariser with a tag and both IVF shapes).

Everything tools/summarize_profile_r02.py writes (tag r03: whole-bench kernel stats, per-mode rows, the two PMC
summaries, the JSON line under the tracer), plus
  r03_batched_only_kernel_stats.csv  kernel stats of a run of ONLY the headline loop (25 steps): AverageNs of
                                     scan_i8_mq_kernel there is what roofline_batched.avg_launch_ms must agree with
  r03_ivf_kernel_stats.csv           kernel stats of ONLY the IVF leg at the BASELINE configs[2] shape (tools/ivf_shard_one.py c3 none)
  r03_ivf_pmc_summary.json           PMC of the two kernels that bound that leg: scan_query_kernel (head) and scan_i8_mq_narrow_kernel
                                     (partition-major second phase): HBM bytes (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024,
                                     MI355X_MICROARCH.md), LDS pipe, VALU issue, wave wait share
  r03_bench_plain.json, r03_shard_sizes.txt, r03_ivf_shard_sizes.txt
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, TAG = os.path.join(ROOT, "gpurun_out"), (sys.argv[1] if len(sys.argv) > 1 else "r04")
P = os.environ.get("QADC_PROFILES_OUT", os.path.join(ROOT, "profiles"))   # (on the GPU box: a small directory under gpurun_out/ that travels back)
os.makedirs(P, exist_ok=True)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profile_r02.py"), TAG], stdout=subprocess.DEVNULL, env=dict(os.environ))


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern), recursive=True), key=os.path.getmtime)
    assert f, pattern
    return f[-1]


shutil.copy(one("prof_%s_batched_kt/**/*_kernel_stats.csv" % TAG), os.path.join(P, "%s_batched_only_kernel_stats.csv" % TAG))
shutil.copy(one("prof_%s_ivf_kt/**/*_kernel_stats.csv" % TAG), os.path.join(P, "%s_ivf_kernel_stats.csv" % TAG))
try:
    shutil.copy(one("prof_%s_ivfc5_kt/**/*_kernel_stats.csv" % TAG), os.path.join(P, "%s_ivfc5_kernel_stats.csv" % TAG))
except AssertionError:
    pass
open(os.path.join(P, "%s_batched_only_bench.json" % TAG), "w").write(
    [l for l in open(os.path.join(G, "prof_%s_batched_kt.log" % TAG)) if l.startswith("{")][-1])
for name in ("bench_plain.json", "shard_sizes.txt", "ivf_shard_sizes.txt"):
    src = os.path.join(G, "%s_%s" % (TAG, name))
    if os.path.exists(src):
        if name.endswith(".json"):
            lines = [l for l in open(src) if l.startswith("{")]
            open(os.path.join(P, "%s_%s" % (TAG, name)), "w").write(lines[-1])
        else:
            shutil.copy(src, os.path.join(P, "%s_%s" % (TAG, name)))


X = 'ivf'


def counters(kind, names):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(one("prof_%s_%s_%s/**/*counter_collection.csv" % (TAG, X, kind)))):
        for n in names:
            if n in r["Kernel_Name"]:
                d = per[(n, int(r["Dispatch_Id"]))]
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


def shape_summary(X_, SHAPE):
    global X
    X = X_
    leg = json.loads([l for l in open(os.path.join(G, "prof_%s_%s_kt.log" % (TAG, X))) if l.startswith("{")][-1])
    names = {"scan_query_kernel": "head (front + first probes of every query, one workgroup per query)",
             "scan_i8_mq_narrow_kernel": "partition-major second phase (8 or 4 queries per pass)",
             "replay_heap_wave_kernel": "wave-per-query heap replay", "order_cands_kernel": "scan order per query"}
    out = {"command": "python3 tools/ivf_shard_one.py %s none  (bench.py's IVF leg of that shape alone; 1024- and 2048-query batches)" % SHAPE,
           "us_per_query_in_the_kernel_trace_run": leg["us_per_query"], "kernels": {}}
    f, w, sq, sq2 = (counters(k, names) for k in ("FETCH_SIZE", "WRITE_SIZE", "sq", "sq2"))
    for n, what in names.items():
        fd = [v for (k, _), v in f.items() if k == n]
        wd = [v for (k, _), v in w.items() if k == n]
        sd = [v for (k, _), v in sq.items() if k == n]
        s2 = [v for (k, _), v in sq2.items() if k == n]
        if not fd or not sd:
            continue
        # 1024-query launches only (the run also has 2048-query batches: the longer half)
        med = sorted(d["_ns"] for d in sd)[len(sd) // 2]
        pick = lambda ds: [d for d in ds if d["_ns"] <= med * 1.4] or ds
        fd, wd, sd, s2 = pick(fd), pick(wd), pick(sd), pick(s2)
        avg = lambda ds, k: sum(d.get(k, 0.0) for d in ds) / max(len(ds), 1)
        cyc = avg(sd, "GRBM_GUI_ACTIVE") / 8
        ent = {"what": what, "launches_averaged": len(sd), "avg_duration_ms_under_pmc": avg(sd, "_ns") / 1e6,
               "hbm_read_bytes_per_launch(FETCH_SIZE*1024*2)": avg(fd, "FETCH_SIZE") * 2048, "hbm_write_bytes_per_launch": avg(wd, "WRITE_SIZE") * 1024,
               "hbm_GBps_under_pmc": (avg(fd, "FETCH_SIZE") * 2048 + avg(wd, "WRITE_SIZE") * 1024) / max(avg(fd, "_ns"), 1),
               "effective_clock_GHz": cyc / max(avg(sd, "_ns"), 1),
               "SQ_LDS_IDX_ACTIVE_per_launch": avg(sd, "SQ_LDS_IDX_ACTIVE"),
               "lds_busy_fraction(SQ_LDS_IDX_ACTIVE/(256*cycles))": avg(sd, "SQ_LDS_IDX_ACTIVE") / (256 * cyc) if cyc else None,
               "lds_conflict_fraction": avg(sd, "SQ_LDS_BANK_CONFLICT") / max(avg(sd, "SQ_LDS_IDX_ACTIVE"), 1),
               "valu_issue_fraction(SQ_INSTS_VALU/(256*cycles))": avg(sd, "SQ_INSTS_VALU") / (256 * cyc) if cyc else None}
        if s2:
            wc = avg(s2, "SQ_WAVE_CYCLES")
            ent["waves_waiting_fraction(SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = avg(s2, "SQ_WAIT_ANY") / wc if wc else None
            ent["waves_waiting_on_lds_fraction(SQ_WAIT_INST_LDS/SQ_WAVE_CYCLES)"] = avg(s2, "SQ_WAIT_INST_LDS") / wc if wc else None
        out["kernels"][n] = ent
    json.dump(out, open(os.path.join(P, "%s_%s_pmc_summary.json" % (TAG, X)), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


for X, SHAPE in (("ivf", "c3"), ("ivfc5", "c5")):
    try:
        shape_summary(X, SHAPE)
    except AssertionError as e:
        print("skipped", X, e)
