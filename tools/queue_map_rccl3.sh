R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for shape in c3 c5; do
  for pads in "" H,H, H,H,H, H,H,H,H, H,H,D, D,H,H,H,; do
    echo -n "$shape torch_before order=${pads}S,C,O,F,W,L,M0: "
    QADC_STREAM_ORDER=${pads}S,C,O,F,W,L,M0 QADC_PROBE_RCCL=torch_before timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P"
  done
done
