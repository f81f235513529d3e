# One of 8 ranks' IVF batch with the merge stream at the lowest (default) / normal priority.  -> gpurun_out/merge_prio_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/merge_prio_ab.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for prio in 0 1; do
  for shape in c3 c5; do
    echo -n "merge_prio $prio $shape range: " >> $OUT
    QADC_MERGE_PRIO=$prio timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
