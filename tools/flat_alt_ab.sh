# Level-path batches alternating between two scan streams (option alt_scan), flat list: 125M-code shard, one of 8 ranks
# (loopback), the 1B list.   -> gpurun_out/flat_alt.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/flat_alt.txt
: > $OUT
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0 QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s" % (j["ms_per_step"], j["value"]))'
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  echo -n "alt_scan=$1 w_low=$2 | 125M single-GPU loop: " >> $OUT
  QADC_W_LOW=$2 QADC_BENCH_OPTS=alt_scan=$1 QADC_BENCH_CODES=125e6 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT
  echo -n "alt_scan=$1 w_low=$2 | rank 0 of 8 (loopback): " >> $OUT
  QADC_W_LOW=$2 QADC_BENCH_OPTS=alt_scan=$1 QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT
  echo -n "alt_scan=$1 w_low=$2 | 1B: " >> $OUT
  QADC_W_LOW=$2 QADC_BENCH_OPTS=alt_scan=$1 python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P" >> $OUT
done
done
cat $OUT
