# round 5, auto layout: one of 8 ranks' IVF batch with the layout probe + pad search on (default) / off, in a fresh process and in one
# that created a torch "nccl" group first.  -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048   layout: %s" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["stream_layout"]))'
for rep in 1; do
for shape in c3 c5; do
for hist in none torch_before; do
for auto in 1; do
  echo -n "$shape history=$hist autolayout=$auto: "
  QADC_STREAM_AUTOLAYOUT=$auto QADC_PROBE_RCCL=$hist timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P"
done
done
done
done
