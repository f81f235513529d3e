# Workgroup counts of the multi-query launches vs the chip's resident capacity (16x4: 67 VGPRs -> 7 workgroups per CU = 1792):
# the 125M-code shard loop under different (mq_min_wgs, mq_codes_per_wg)
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step" % j["ms_per_step"])'
for o in "mq_min_wgs=4096,mq_codes_per_wg=65536" "mq_min_wgs=3584,mq_codes_per_wg=65536" "mq_min_wgs=3584,mq_codes_per_wg=69400" "mq_min_wgs=3584,mq_codes_per_wg=104100" "mq_min_wgs=5376,mq_codes_per_wg=69400" "mq_min_wgs=1792,mq_codes_per_wg=104100" "mq_min_wgs=7168,mq_codes_per_wg=52050" "mq_min_wgs=4096,mq_codes_per_wg=65536"; do
echo -n "$o  ${CODES:-125e6}: "; QADC_BENCH_OPTS=$o QADC_BENCH_CODES=${CODES:-125e6} python3 bench.py --steps ${STEPS:-60} --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done
