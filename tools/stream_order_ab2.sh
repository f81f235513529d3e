# Round 4, second pass: (1) predictions of the "scan stream's pipe" reading of stream_order_ab.sh; (2) the candidate order under
# three process histories; (3) bench.py's own IVF leg against the stand-alone tool on the SAME box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stream_order2.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
run() { # shape place order wgq hist
  echo -n "$1 $2 order=$3 wgq_stream=$4 hist=$5: " >> $OUT
  QADC_PROBE_HISTORY=$5 QADC_STREAM_ORDER=$3 QADC_WGQ_STREAM=$4 timeout 300 python3 $R/tools/ivf_shard_one.py $1 $2 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
}
for shape in c3 c5; do
  for order in "S,C,O,F,W,L,M0" "H,S,C,O,F,W,L,M0" "S,C,O,F,H,W,L,M0" "S,L,C,O,F,W,M0" "S,C,O,L,F,W,M0" "S,C,L,F,O,W,M0" "S,C,O,F,W,L,M0,M1,M2" "S,C,O,F,W,L,M0,M1,N,M2" "S,C,O,F,W,L,M0,M1"; do
    run $shape range $order 0 fresh
  done
  for hist in destroyed alive; do
    for order in "S,C,O,F,W,L,M0" "S,C,O,F,W,L,M0,M1,N,M2"; do
      run $shape range $order 0 $hist
      run $shape range $order 1 $hist
    done
  done
  run $shape range "S,C,O,F,W,L,M0,M1,N,M2" 1 fresh
  run $shape none "S,C,O,F,W,L,M0,M1,N,M2" 0 fresh
done
echo "--- bench.py's IVF legs in the bench process (default order of this build)" >> $OUT
QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_C2=0 python3 $R/bench.py --steps 3 --warmup 1 2>/dev/null | python3 -c '
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print("bench.py ivf: %.3f us/query (%.3f at 2048); c5: %.3f (%.3f)" % (j["ivf"]["us_per_query"], j["ivf"]["us_per_query_at_2048_query_batches"], j["ivf_c5_one_gpu"]["us_per_query"], j["ivf_c5_one_gpu"]["us_per_query_at_2048_query_batches"]))' >> $OUT 2>&1
run c3 none "S,M0,M1,M2,W,C,O,F,L" 0 fresh
cat $OUT
