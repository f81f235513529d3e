# One of 8 ranks' flat step (loopback) against batches in flight (QADC_BENCH_LEAD).  -> gpurun_out/flat_lead_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/flat_lead_sweep.txt
: > $OUT
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step" % j["ms_per_step"])'
for rep in 1 2; do
for lead in 2 3 4 5 7; do
  echo -n "lead $lead: " >> $OUT
  QADC_BENCH_LEAD=$lead QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 $R/bench.py --steps 60 --warmup 8 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT 2>&1
done
done
cat $OUT
