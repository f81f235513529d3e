# Kernel timeline of the 10M-code flat step loop (tools/flat_small_breakdown.py): tools/flat_small_trace.sh [options]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/smalltl -- python3 $R/tools/flat_small_breakdown.py "$@" > $R/gpurun_out/smalltl.log 2>&1
python3 $R/tools/ivf_shard_timeline.py $R/gpurun_out/smalltl 1.2 150 sort_cands > $R/gpurun_out/smalltl.txt 2>&1
