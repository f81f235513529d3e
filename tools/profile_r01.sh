#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: bash tools/profile_r01.sh [tag]   -> gpurun_out/prof_<tag>_{kt,fetch,write,lds}
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -- python3 $R/bench.py --steps 5 --warmup 1 > $R/gpurun_out/prof_${TAG}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_lds -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}_lds.log 2>&1
grep -h '^{' $R/gpurun_out/prof_${TAG}_kt.log | cut -c1-400
