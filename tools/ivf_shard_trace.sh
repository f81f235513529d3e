# Kernel timeline of one rank of an 8-rank IVF leg (loopback merge): tools/ivf_shard_trace.sh c5|c3 [range|whole]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ivftl -- python3 $R/tools/ivf_shard_one.py ${1:-c5} ${2:-range} 0 > $R/gpurun_out/ivftl.log 2>&1
python3 $R/tools/ivf_shard_timeline.py $R/gpurun_out/ivftl > $R/gpurun_out/ivftl.txt 2>&1
