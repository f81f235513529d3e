# option plan_early (tables, state clear, partition-major plan off the scan stream) on / off, same process layout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/plan_early.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048  fb %d" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["of_them_redone_on_the_level_path"]))'
for rep in 1 2; do
for pe in 1 0; do
  for shape in c3 c5; do
    for place in none range; do
      echo -n "plan_early=$pe $shape $place: " >> $OUT
      QADC_BENCH_IVF_OPTS=plan_early=$pe timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
    done
  done
done
done
cat $OUT
