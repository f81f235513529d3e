# the hardware-queue map and the one-of-8 IVF batch with a LIVE RCCL communicator in the process
# (world 1 is all a 1-GPU box allows).  -> gpurun_out/r05_queue_map_rccl.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_queue_map_rccl.txt
: > $OUT
export TMPDIR=/tmp QADC_TEST_HOOKS=1
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for shape in c3 c5; do
for mode in none lib_after torch_before lib_after_torch_before; do
  echo -n "$shape one of 8 ranks (range split, loopback merge), live RCCL communicator: $mode: " >> $OUT
  QADC_PROBE_RCCL=$mode timeout 300 python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
done
done
done
cd /tmp
for mode in lib_after torch_before; do
  rm -rf /tmp/qm_$mode
  QADC_PROBE_RCCL=$mode rocprofv3 --kernel-trace --output-format csv -d /tmp/qm_$mode -- python3 $R/tools/ivf_shard_one.py c5 range 0 > /dev/null 2>&1
  echo "--- hardware-queue map (rocprofv3 --kernel-trace, tools/queue_map.py), c5 one of 8 ranks, live RCCL communicator: $mode" >> $OUT
  python3 $R/tools/queue_map.py /tmp/qm_$mode >> $OUT 2>&1
done
cat $OUT
