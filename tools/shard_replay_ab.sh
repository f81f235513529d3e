R=${GRAFT_REPO_ROOT:-$(pwd)}
export QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for shape in c3 c5; do
for opt in "dist_shard_replay=0" "dist_shard_replay=1" "dist_shard_replay=1,dist_share_lag=2" "dist_shard_replay=1,dist_share_lag=0"; do
  echo -n "$shape one of 8 ranks, $opt: "
  QADC_BENCH_IVF_OPTS=$opt timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>&1 | python3 -c "$P"
done
done
done
