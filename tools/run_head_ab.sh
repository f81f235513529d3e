# round 6: the full GPU suite with the 512-thread head as the 16x4 default, then one of 8 ranks' IVF batch (loopback stand-in), 8 vs 16 waves
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_gputest_head.txt
cat gpurun_out/r06_gputest_head.txt
RANKS_EMU=0 python3 tools/ivf_shard_sizes.py c3 > gpurun_out/r06_ivf_shard_wg512.txt 2>/dev/null
QADC_BENCH_IVF_OPTS=head_wg=0,wgq_group_head=3 RANKS_EMU=0 python3 tools/ivf_shard_sizes.py c3 > gpurun_out/r06_ivf_shard_wg1024.txt 2>/dev/null
echo "--- 8 waves (default)"; cat gpurun_out/r06_ivf_shard_wg512.txt; echo "--- 16 waves"; cat gpurun_out/r06_ivf_shard_wg1024.txt
