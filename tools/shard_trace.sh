# kernel trace of the per-rank step at 1/8 of the headline list (125M codes x 32 queries); FORCE_DIST=1 for the multi-rank loop
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export QADC_BENCH_CODES=125e6 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
rm -rf gpurun_out/shard_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/shard_trace -o st --output-format csv -- python3 bench.py --steps 60 --warmup 5 > gpurun_out/shard_trace.log 2>&1
grep '^{' gpurun_out/shard_trace.log | cut -c1-200
