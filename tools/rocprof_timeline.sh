# Kernel timeline of pipelined batches under rocprofv3 (run on the GPU box): bash tools/rocprof_timeline.sh [codes] [queries]; then python tools/timeline.py gpurun_out/tl
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
N=${1:-125e6} NQ=${2:-32} STEPS=12 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/tools/pipeline_probe.py > $R/gpurun_out/tl.log 2>&1
