R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
PI='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_W_PRIO=high QADC_MERGE_PRIO2=high QADC_WGQ_STREAM=1
for order in W,C,O,F,S,L,M0 W,S,C,O,F,L,M0 S,W,C,O,F,L,M0; do
  for shape in c3 c5; do
    for hist in none torch_before; do
      echo -n "[order $order, W high, IVF on W, merge high] IVF $shape one of 8 ranks, history=$hist: "; QADC_STREAM_ORDER=$order QADC_PROBE_RCCL=$hist python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$PI"
    done
  done
done
