# Kernel timeline of one of 8 ranks' FLAT step loop (loopback merge): tools/flat_shard_trace.sh [dist opts]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
export QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 QADC_BENCH_DIST_OPTS=$1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/flattl -- python3 $R/bench.py --steps 40 --warmup 5 > $R/gpurun_out/flattl.log 2>&1
python3 $R/tools/ivf_shard_timeline.py $R/gpurun_out/flattl 4.0 30 sort_cands > $R/gpurun_out/flattl.txt 2>&1
