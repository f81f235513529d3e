# Round 5 experiment: the next batch's head BESIDE the partition-major phase (option group_stream: phase + ordering on the level
# path's scan stream — lowest priority, same pipe as the query-kernel stream, whose waiting head dispatch therefore goes first)
# with one head workgroup per CU (head_lds_pad) or two.  -> gpurun_out/group_stream_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/group_stream_ab.txt
: > $OUT
cd $R
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for opts in "" "group_stream=1" "group_stream=1,head_lds_pad=20000" "head_lds_pad=20000"; do
  for sh in c3 c5; do
    echo -n "[$opts] $sh none: " >> $OUT
    QADC_BENCH_IVF_OPTS="$opts" timeout 300 python3 tools/ivf_shard_one.py $sh none 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cat $OUT
