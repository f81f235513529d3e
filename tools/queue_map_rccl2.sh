# second pass of tools/queue_map_rccl.sh: what about a communicator created before the library's stream set disturbs the layout, and
# which remedies recover it (the library's stream set created first; idle pad streams in front of the set)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_queue_map_rccl2.txt
: > $OUT
export TMPDIR=/tmp QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
run() { # shape mode order
  echo -n "$1 rccl=$2 order=${3:-default}: " >> $OUT
  if [ -n "$3" ]; then export QADC_STREAM_ORDER=$3; else unset QADC_STREAM_ORDER; fi
  QADC_PROBE_RCCL=$2 timeout 300 python3 $R/tools/ivf_shard_one.py $1 range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
}
for shape in c3 c5; do
  run $shape none
  run $shape torch_cuda_only
  run $shape set_first
  run $shape torch_before
  for pads in D D,D D,D,D H H,H N N,N H,D H,H,D,D; do
    run $shape torch_before $pads,S,C,O,F,W,L,M0
  done
  run $shape torch_before S,W,C,L,O,M0,F
  run $shape torch_before S,C,O,F,L,M0
done
cat $OUT
