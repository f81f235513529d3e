# Head length x ramp growth, one of 8 ranks (loopback) and one GPU.   -> gpurun_out/head_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_sweep.txt
: > $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048  fallbacks %d" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["of_them_redone_on_the_level_path"]))'
for shape in c5 c3; do
  for place in range none; do
    for ramp in 1 2 3; do
      for head in 2 3 4; do
        [ $place = none ] && [ $head = 4 ] && continue
        if [ $place = range ]; then opts="wgq_ramp_shift=$ramp,wgq_group_head_dist=$head"; else opts="wgq_ramp_shift=$ramp,wgq_group_head=$head"; fi
        echo -n "$shape $place $opts: " >> $OUT
        QADC_BENCH_IVF_OPTS=$opts timeout 300 python3 $R/tools/ivf_shard_one.py $shape $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
      done
    done
  done
done
cat $OUT
