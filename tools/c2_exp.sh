# BASELINE configs[1] (flat 10M): the one-query-per-pass mode with the lone run through the multi-query kernel's 4-seat form
# (option mq_single), at several depths; and the same option on the 1B list's roofline leg.   -> gpurun_out/c2_exp.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c2_exp.txt
: > $OUT
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=j["c2"]; r=j["roofline_c2"]; print("c2 batched %.3f ms/step | one query %.4f ms (%.0f GB/s wall; launches avg %.1f us, %.0f GB/s inside) | 1B one query: frac %s, %.3f ms/query" % (c["batched"]["ms_per_step"], c["one_query_per_pass"]["ms_per_query"], r["achieved"], r["streaming_launches"]["avg_launch_ms"]*1e3, r["streaming_launches"]["GBps_inside_the_launches"], j["roofline"].get("frac"), j["roofline"].get("ms_per_query_wall") or 0))'
for opts in "" "mq_single=1" ; do
  for depth in 3 8; do
    echo -n "[c2 opts=$opts depth=$depth] " >> $OUT
    QADC_BENCH_C2_OPTS=$opts QADC_BENCH_C2_DEPTH=$depth QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_CODES=1e7 python3 $R/bench.py --steps 3 --warmup 1 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
for opts in "" "mq_single=2"; do
  echo -n "[1B opts=$opts] " >> $OUT
  QADC_BENCH_OPTS=$opts QADC_BENCH_C2=1 QADC_BENCH_SINGLE_QUERIES=48 python3 $R/bench.py --steps 3 --warmup 1 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
done
cat $OUT
