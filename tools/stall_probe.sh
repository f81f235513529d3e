# multi-rank loop (one rank, RCCL world 1) on the 1/8 shard: per-iteration host times with option sets
export QADC_BENCH_CODES=125e6 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_STEP_LOG=gpurun_out/steplog.txt
rm -f gpurun_out/steplog.txt
P='import sys,json; j=json.loads(sys.stdin.read()); print("mean %.4f ms/step" % (j["ms_per_step"]))'
for rep in 1 2 3; do
for o in "$@"; do echo -n "[$o] " >> gpurun_out/steplog.txt; QADC_BENCH_OPTS="$o" python3 bench.py --steps 200 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P" >> gpurun_out/steplog.txt; done; done
grep -v "^steps 5:" gpurun_out/steplog.txt
