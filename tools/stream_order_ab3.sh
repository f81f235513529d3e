# Round 4, third pass: the adopted stream order under process histories (an index created and destroyed first: streams destroyed
# in reverse creation order — the default — or forward), bench.py's IVF / c2 legs in the bench process, the flat one-of-8 step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stream_order3.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
run() { # shape place hist forward
  echo -n "$1 $2 hist=$3 destroy_forward=$4: " >> $OUT
  QADC_PROBE_HISTORY=$3 QADC_DESTROY_FORWARD=$4 timeout 300 python3 $R/tools/ivf_shard_one.py $1 $2 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
}
for rep in 1 2 3; do
for shape in c3 c5; do
  run $shape range fresh 0
  run $shape range destroyed 0
  [ $rep = 1 ] && run $shape range destroyed 1
  [ $rep = 1 ] && run $shape range alive 0
  [ $rep = 1 ] && run $shape none fresh 0
  [ $rep = 1 ] && run $shape none destroyed 0
done
done
echo "--- bench.py's IVF + c2 legs in the bench process" >> $OUT
QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_LATENCY=0 python3 $R/bench.py --steps 3 --warmup 1 2>/dev/null | python3 -c '
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print("bench.py ivf: %.3f us/query (%.3f at 2048); c5: %.3f (%.3f); c2 batched %.3f ms/step, one query %.3f ms" % (j["ivf"]["us_per_query"], j["ivf"]["us_per_query_at_2048_query_batches"], j["ivf_c5_one_gpu"]["us_per_query"], j["ivf_c5_one_gpu"]["us_per_query_at_2048_query_batches"], j["c2"]["batched"]["ms_per_step"], j["c2"]["one_query_per_pass"]["ms_per_query"]))' >> $OUT 2>&1
echo "--- flat, one of 8 ranks (tools/dist_sizes3.sh)" >> $OUT
bash $R/tools/dist_sizes3.sh >> $OUT 2>&1
cat $OUT
