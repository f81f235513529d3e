# Two merge streams, the second placed on a chosen pipe by idle pad streams (QADC_STREAM_ORDER), against the default single one.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/merge2.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for cfg in "1 S,C,O,F,W,L,M0" "2 S,C,O,F,W,L,M0,N,N,M1" "2 S,C,O,F,W,L,M0,N,N,D,M1" "2 S,C,O,F,W,L,M0,M1" "3 S,C,O,F,W,L,M0,N,N,M1,D,N,M2"; do
  set -- $cfg
  for shape in c3 c5; do
    echo -n "merge_streams=$1 order=$2 $shape range: " >> $OUT
    QADC_MERGE_STREAMS=$1 QADC_STREAM_ORDER=$2 timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
