#!/bin/bash
# Kernel trace of the IVF leg and the single-query latency leg of bench.py only (GPU box, via gpurun).
TAG=${1:-r02_ivf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_BENCH_CODES=2e7 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_32X4=0 QADC_BENCH_PMC=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof_${TAG}.log 2>&1
head -14 $R/gpurun_out/prof_${TAG}/*/*_kernel_stats.csv | cut -c1-60,150-330
grep -h '^{' $R/gpurun_out/prof_${TAG}.log | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(j['ivf'])[:600]); print(json.dumps(j['latency_us_single_query'])[:300])"
