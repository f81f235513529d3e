# round 5: the device replay of a partition-major batch deferred behind the NEXT batch's head launch (option replay_defer) on / off,
# one GPU, bench.py's IVF legs (1024- and 2048-query batches, four in flight).  -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=j.get("roofline") or {}; print("%.3f ms/batch  %.3f us/q  %.3f us/q@2048 | head %.3f ms (pipelined) grouped %.3f" % (j["ms_per_batch"], j["us_per_query"], j["us_per_query_at_2048_query_batches"], r.get("head",{}).get("avg_launch_ms",0), r.get("avg_launch_ms",0)))'
for rep in 1 2; do
for shape in c3 c5; do
for opt in replay_defer=0 replay_defer=1; do
  echo -n "$shape one GPU, $opt: "
  QADC_BENCH_IVF_OPTS=$opt timeout 300 python3 $R/tools/ivf_shard_one.py $shape none 2>&1 | python3 -c "$P"
done
done
done
