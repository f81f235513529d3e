# Same-box comparison of several builds (quick-adc_amd/libqadc_hip*.so) on one of 8 ranks' IVF batch.  -> gpurun_out/rank_ab_libs.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/rank_ab_libs.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for lib in $(cd $R/quick-adc_amd && ls libqadc_hip*.so); do
  for shape in c3 c5; do
      echo -n "$lib $shape range: " >> $OUT
      QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
