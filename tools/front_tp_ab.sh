# round 5: the throughput front of partition-major batches (option front_tp) on / off — bench.py's IVF legs on one GPU and one of 8 ranks
R=${GRAFT_REPO_ROOT:-$(pwd)}
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); r=j.get("roofline") or {}; print("%.3f ms/batch  %.3f us/q  %.3f us/q@2048 | head %.3f ms (pipelined) grouped %.3f" % (j["ms_per_batch"], j["us_per_query"], j["us_per_query_at_2048_query_batches"], r.get("head",{}).get("avg_launch_ms",0), r.get("avg_launch_ms",0)))'
for rep in 1 2; do
for shape in c3 c5; do
for opt in front_tp=0 front_tp=1; do
  echo -n "$shape one GPU, $opt: "
  QADC_BENCH_IVF_OPTS=$opt timeout 300 python3 $R/tools/ivf_shard_one.py $shape none 2>&1 | python3 -c "$P"
done
done
done
