# experiments on the per-rank step at 1/8 of the headline list
export QADC_BENCH_CODES=${CODES:-125e6} QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s" % (j["ms_per_step"], j["value"]))'
for rep in 1 2; do
for o in "$@"; do
echo -n "[$o] "; QADC_BENCH_OPTS="$o" python3 bench.py --steps ${STEPS:-100} --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done; done
