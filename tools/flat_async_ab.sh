export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
export QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step" % j["ms_per_step"])'
for o in ${OPTS:-dist_async=1 dist_async=0}; do
 for i in 1 2; do
  rm -f gpurun_out/steplog.txt
  echo -n "$o: "; QADC_BENCH_STEP_LOG=gpurun_out/steplog.txt QADC_BENCH_DIST_OPTS=$o python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
  grep "host ms" gpurun_out/steplog.txt | tail -1
 done
done
