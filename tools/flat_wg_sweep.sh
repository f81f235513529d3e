# Flat 8-queries-per-pass loop against codes per workgroup (option mq_codes_per_wg): 125 M-code shard and the 1B list.  -> gpurun_out/flat_wg_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/flat_wg_sweep.txt
: > $OUT
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s" % (j["ms_per_step"], j["value"]))'
for rep in 1 2; do
for v in 65536 32768 16384; do
  echo -n "mq_codes_per_wg $v 125M: " >> $OUT
  QADC_BENCH_OPTS=mq_codes_per_wg=$v QADC_BENCH_CODES=125e6 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT 2>&1
  echo -n "mq_codes_per_wg $v 1B:   " >> $OUT
  QADC_BENCH_OPTS=mq_codes_per_wg=$v python3 $R/bench.py --steps 15 --warmup 3 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT 2>&1
done
done
cat $OUT
