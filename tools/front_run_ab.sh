# Early bound levels on the front stream (front_run_max) or everything on the main stream, by list size (bench.py's flat loop)
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step" % j["ms_per_step"])'
for n in ${SIZES:-1e7 3e7 6e7}; do for i in 1 2; do for o in "front_min_batch=0" "front_run_max=0"; do
echo -n "N=$n [$o]: "; QADC_BENCH_OPTS=$o QADC_BENCH_CODES=$n python3 bench.py --steps 200 --warmup 10 2>/dev/null | grep "^{" | python3 -c "$P"
done; done; done
