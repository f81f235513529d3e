# Does the SET / ORDER of streams an index creates change its speed?  (round 4: single-GPU IVF legs and one of 8 ranks)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stream_order.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for shape in c3 c5; do
for place in none range; do
for order in "S,C,O,F,L,M0,M1,M2,W" "S,M0,M1,M2,W,C,O,F,L" "S,C,O,F" "S,C,O,F,W,L,M0" "W,S,C,O,F,L,M0,M1,M2"; do
  for wgq in 0 1; do
    case "$order" in *W*) ;; *) [ $wgq = 1 ] && continue;; esac
    echo -n "$shape $place order=$order wgq_stream=$wgq: " >> $OUT
    QADC_STREAM_ORDER=$order QADC_WGQ_STREAM=$wgq timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
done
cat $OUT
