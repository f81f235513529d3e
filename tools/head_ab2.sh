# Same-box A/B of two builds (libqadc_hip.so vs libqadc_hip_nopipe.so = whatever second build was linked under that name): head
# clocks, the one-GPU legs, one of 8 ranks' batch.   -> gpurun_out/head_ab2.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_ab2.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  for shape in c3 c5; do
    echo -n "$lib $shape: " >> $OUT
    QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_head_cycles.py $shape 2>&1 | tail -1 >> $OUT
    for place in none range; do
      echo -n "$lib $shape $place: " >> $OUT
      QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
    done
  done
done
done
cat $OUT
