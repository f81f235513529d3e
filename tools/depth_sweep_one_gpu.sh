# One GPU's IVF legs against batches in flight.  -> gpurun_out/depth_sweep_one_gpu.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/depth_sweep_one_gpu.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for d in 2 3 4 6 8; do
  for shape in c3 c5; do
    echo -n "depth $d $shape none: " >> $OUT
    QADC_BENCH_IVF_DEPTH=$d timeout 300 python3 $R/tools/ivf_shard_one.py $shape none 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
