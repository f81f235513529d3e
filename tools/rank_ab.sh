# Same-box A/B of two builds on one of 8 ranks' IVF batch (and the one-GPU legs): libqadc_hip.so vs libqadc_hip_nopipe.so.  -> gpurun_out/rank_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/rank_ab.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2 3; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  for shape in c3 c5; do
    for place in ${RANK_AB_PLACES:-range}; do
      echo -n "$lib $shape $place: " >> $OUT
      QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
    done
  done
done
done
cat $OUT
