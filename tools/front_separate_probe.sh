R=${GRAFT_REPO_ROOT:-$(pwd)}
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for shape in c3 c5; do
  echo -n "$shape one GPU (no merge): "; python3 $R/tools/ivf_shard_one.py $shape none 2>&1 | python3 -c "$P"
  echo -n "$shape world 1 loopback merge, front inside the head: "; WORLD_EMU=1 QADC_BENCH_IVF_OPTS=dist_shard_front=0 python3 $R/tools/ivf_shard_one.py $shape range 0 2>&1 | python3 -c "$P"
  echo -n "$shape world 1 loopback merge, front as its own launch on the front stream: "; WORLD_EMU=1 QADC_BENCH_IVF_OPTS=dist_shard_front=2 python3 $R/tools/ivf_shard_one.py $shape range 0 2>&1 | python3 -c "$P"
done
done
