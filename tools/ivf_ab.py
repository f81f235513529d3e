"""Same-box A/B of the IVF legs under option sets: one process, one GPU box, every arm builds its own index.
usage: python tools/ivf_ab.py c3|c5|both "name1:opt=v,opt=v" "name2:..."   (an arm "base:" has no options)
Prints per arm: us per query (1024- and 2048-query batches), the three launches' pipelined and stand-alone times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QADC_BENCH_CPU_SECONDS", "0")
import bench  # noqa: E402


def main():
    import torch
    import pyqadc
    torch.zeros(1, device="cuda:0")
    pyqadc.device_prepare(0)
    import gc                                                  # as bench.py's main(): with torch loaded a full collection takes ~50 ms and
    gc.collect(); gc.freeze(); gc.disable()                    # lands in some arm's timed loop (the "second arm" stall of the round-6 runs)
    shapes = ["c3", "c5"] if sys.argv[1] == "both" else [sys.argv[1]]
    arms = [a.split(":", 1) for a in sys.argv[2:]]
    reps = int(os.environ.get("AB_REPS", 2))
    for shape in shapes:
        for rep in range(reps):
            for name, opts in arms:
                os.environ["QADC_BENCH_IVF_OPTS"] = opts
                kw = {} if shape == "c3" else dict(M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000)
                r = bench.ivf_leg(0, **kw)
                ro = r.get("roofline", {})
                al = ro.get("launches_alone_on_the_gpu") or {}
                print(json.dumps({"shape": shape, "arm": name, "rep": rep, "us_per_query": round(r["us_per_query"], 4),
                                  "us_per_query_2048": round(r["us_per_query_at_2048_query_batches"], 4),
                                  "head_ms_pipelined": round(ro.get("head", {}).get("avg_launch_ms", 0), 4),
                                  "grouped_ms_pipelined": round(ro.get("avg_launch_ms", 0), 4),
                                  "order_ms_pipelined": round(ro.get("order_cands_avg_launch_ms", 0), 4),
                                  "head_ms_alone": round(al.get("head_ms", 0), 4), "grouped_ms_alone": round(al.get("grouped_scan_ms", 0), 4),
                                  "seat_fill": round(ro.get("seat_fill", 0), 4), "grouped_frac_lds_alone": round(al.get("grouped_scan_frac_of_lds", 0), 4),
                                  "head_frac_hbm_alone": round(al.get("head_frac_of_hbm", 0), 4)}), flush=True)


if __name__ == "__main__":
    main()
