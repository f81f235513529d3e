R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048   layout: %s" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["stream_layout"]))'
for shape in c3 c5; do
for hist in none torch_before; do
for sp in low normal high; do
  echo -n "$shape history=$hist scan_prio=$sp: "
  QADC_SCAN_PRIO=$sp QADC_PROBE_RCCL=$hist timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P"
done
echo -n "$shape history=$hist scan_prio=high merge_prio=high: "
QADC_SCAN_PRIO=high QADC_MERGE_PRIO2=high QADC_PROBE_RCCL=$hist timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P"
done
done
echo "--- one GPU (no merge)"
for shape in c3 c5; do for sp in low high; do echo -n "$shape none scan_prio=$sp: "; QADC_SCAN_PRIO=$sp timeout 300 python3 $R/tools/ivf_shard_one.py $shape none 2>/dev/null | python3 -c "$P"; done; done
