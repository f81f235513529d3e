# Same-box A/B of two builds of the query kernel (make -C quick-adc_amd ab): the software-pipelined walk against the plain one.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_ab.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  for shape in c3 c5; do
    echo -n "$lib $shape: " >> $OUT
    QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_head_cycles.py $shape 2>&1 | tail -1 >> $OUT
    echo -n "$lib $shape whole leg, one GPU: " >> $OUT
    QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_shard_one.py $shape none 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
