"""One of 8 ranks' IVF step on ONE GPU, for both placements of SURVEY.md 8e (what tools/dist_sizes2.sh does for the flat
list): the process builds rank r's part of the BASELINE configs[2] / configs[4] databases, submits the full query batches
and collects them through qadc_dist_collect with the loopback stand-in for the other 7 ranks (qadc_dist_init_loopback:
the gather returns this rank's block 8 times, so the merge replays a world's worth of entries).

    python3 tools/ivf_shard_sizes.py [c3] [c5]      -> profiles/r03_ivf_shard_sizes.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
os.environ.setdefault("QADC_BENCH_CPU_SECONDS", "0")   # (no CPU sample for these probes)
import bench  # noqa: E402

WORLD = int(os.environ.get("WORLD_EMU", 8))
RANKS = [int(x) for x in os.environ.get("RANKS_EMU", "0,5").split(",")]
SHAPES = {"c3": dict(M=16, K=4096, MA=32, dim=128, N=int(1e8), seed0=1000),
          "c5": dict(M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000)}


def main():
    which = [a for a in sys.argv[1:] if a in SHAPES] or ["c3", "c5"]
    for name in which:
        kw = SHAPES[name]
        base = bench.ivf_leg(0, **kw)
        print("%s unsharded (one GPU holds everything): %.3f us/query, %.3f ms per 1024-query batch, %.3f us/query at 2048; "
              "1/%d of it = %.3f ms per batch" % (name, base["us_per_query"], base["ms_per_batch"], base["us_per_query_at_2048_query_batches"],
                                                 WORLD, base["ms_per_batch"] / WORLD), flush=True)
        for placement in ("whole", "range"):
            for r in RANKS:
                shard = dict(rank=r, world=WORLD, placement=placement, init=lambda ix, r=r: ix.dist_init_loopback(r, WORLD),
                             merge="loopback stand-in")
                o = bench.ivf_leg(0, shard=shard, **kw)
                print("%s rank %d of %d, %-5s placement: %.3f ms per 1024-query batch (%.3f us/query), %.3f us/query at 2048-query "
                      "batches; %d codes on the rank, %.0f candidates/query from this rank, grouped batches %d (fallbacks %d), host ms/batch %s"
                      % (name, r, WORLD, placement, o["ms_per_batch"], o["us_per_query"], o["us_per_query_at_2048_query_batches"],
                         o["codes_on_this_rank"], o["candidates_per_query_this_rank"], o["batches_through_partition_major_second_phase"],
                         o["of_them_redone_on_the_level_path"], json.dumps({k: round(v, 3) for k, v in o["host_ms_per_batch"].items()})), flush=True)


if __name__ == "__main__":
    main()
