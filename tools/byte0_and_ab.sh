# Round 5: byte 0's nibble extracts as plain v_and_b32 (full rate) instead of SDWA (half rate) in the multi-query scan bodies
# (QADC_BYTE0_PLAIN_AND; second build: make ab AB_QUERY_FLAGS= AB_KERNEL_FLAGS=-DQADC_BYTE0_PLAIN_AND=0): the IVF legs through
# tools/head_ab2.sh and the flat 1B steps (one query per pass + batched) of bench.py under both libraries.  -> gpurun_out/byte0_and_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/byte0_and_ab.txt
cd $R
bash tools/head_ab2.sh > /dev/null 2>&1
cp gpurun_out/head_ab2.txt $OUT
OFF="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_32X4=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_C2=0"
for rep in 1 2; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  echo -n "$lib flat 1B: " >> $OUT
  env $OFF QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 600 python3 bench.py --steps 20 --warmup 2 2>/dev/null | python3 -c 'import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("one query per pass %.3f ms/step  batched %.3f ms/step" % (j["ms_per_step"], j["ms_per_step_batched"]))' >> $OUT 2>&1
done
done
cat $OUT
