# One of 8 ranks' IVF batch (loopback merge, tools/ivf_shard_one.py) under the stream plans of round 4, in three process
# histories: fresh | an index created and destroyed first | another index alive.   -> gpurun_out/stream_plan.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stream_plan.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048  grouped %d fb %d" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["batches_through_partition_major_second_phase"], j["of_them_redone_on_the_level_path"]))'
export QADC_TEST_HOOKS=1
for shape in ${SHAPES:-c5 c3}; do
for hist in ${HISTS:-fresh destroyed alive}; do
for plan in ${PLANS:-"0 3" "1 3" "1 1" "0 1"}; do
  set -- $plan
  for rep in 1 2; do
    echo -n "$shape hist=$hist wgq_stream=$1 merge_streams=$2 rep$rep: " >> $OUT
    QADC_PROBE_HISTORY=$hist QADC_WGQ_STREAM=$1 QADC_MERGE_STREAMS=$2 timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
done
cat $OUT
