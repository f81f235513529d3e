# One of 8 ranks' IVF batch against the number of batches in flight (bench.ivf_leg, loopback merge).  -> gpurun_out/depth_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/depth_sweep.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for d in 3 4 5 6 8; do
  for shape in c3 c5; do
    echo -n "depth $d $shape range: " >> $OUT
    QADC_BENCH_IVF_DEPTH=$d timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
