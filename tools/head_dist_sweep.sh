# One of 8 ranks' IVF batch against the head's length under the merge (option wgq_group_head_dist).  -> gpurun_out/head_dist_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_dist_sweep.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048  fallbacks %s" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j.get("of_them_redone_on_the_level_path")))'
for rep in 1 2; do
for v in ${HEAD_DIST_VALUES:-2 3 4 5 6}; do
  for shape in c3 c5; do
    echo -n "head_dist $v $shape range: " >> $OUT
    QADC_BENCH_IVF_OPTS=wgq_group_head_dist=$v timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
