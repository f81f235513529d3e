# bench.ivf_leg (pipelined batches) with the partition-major phase at 64 K / 16 K / 8 K codes per workgroup.  -> gpurun_out/group_wg_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/group_wg_ab.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for v in 65536 16384 8192; do
  for shape in c3 c5; do
    for place in none range; do
      echo -n "codes_per_wg $v $shape $place: " >> $OUT
      QADC_BENCH_IVF_OPTS=wgq_group_codes_per_wg=$v timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
    done
  done
done
done
cat $OUT
