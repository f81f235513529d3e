# Same-box A/B of two builds, head clocks only (tools/ivf_head_cycles.py): libqadc_hip.so vs libqadc_hip_nopipe.so.  -> gpurun_out/head_ab3.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_ab3.txt
: > $OUT
export QADC_TEST_HOOKS=1
for rep in 1 2 3; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  for shape in c3 c5; do
    echo -n "$lib " >> $OUT
    QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_head_cycles.py $shape 2>&1 | tail -1 >> $OUT
  done
done
done
cat $OUT
