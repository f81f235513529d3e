#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command + separate PMC
# passes of BOTH scan modes (one query per pass = bench.py --pmc-leg; batched = bench.py's headline loop only).
# Usage: bash tools/profile_r02.sh [tag]   -> gpurun_out/prof_<tag>_{kt,single_*,batched_*}
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
# (1) the default command, every leg, under the kernel trace (the in-run PMC children are skipped under a profiler)
QADC_BENCH_CPU_SECONDS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -- python3 $R/bench.py --steps 5 --warmup 1 > $R/gpurun_out/prof_${TAG}_kt.log 2>&1
# (2) one query per pass: HBM counters (separate passes) and the SQ/GRBM set
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_${TAG}_single_$C -- python3 $R/bench.py --pmc-leg > $R/gpurun_out/prof_${TAG}_single_$C.log 2>&1
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_single_sq -- python3 $R/bench.py --pmc-leg > $R/gpurun_out/prof_${TAG}_single_sq.log 2>&1
# (3) batched mode (32 queries per step, 8 per pass): the headline loop only
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_32X4=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_${TAG}_batched_$C -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}_batched_$C.log 2>&1
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_batched_sq -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}_batched_sq.log 2>&1
grep -h '^{' $R/gpurun_out/prof_${TAG}_kt.log | cut -c1-600
