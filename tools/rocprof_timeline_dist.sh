# Kernel timeline of the multi-rank step loop (tools/dist_probe.py) under rocprofv3; then python tools/timeline.py gpurun_out/tld
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
N=${1:-5e8} WORLD_EMU=${2:-2} STEPS=${3:-24} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tld -- python3 $R/tools/dist_probe.py > $R/gpurun_out/tld.log 2>&1
