# step time of one of 8 ranks' shard (125M codes) through: single-GPU loop, multi-rank loop native merge, torch merge
export QADC_BENCH_CODES=125e6 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s  %s" % (j["ms_per_step"], j["value"], j.get("multi_gpu_merge")))'
for i in 1 2; do
echo -n "single-GPU loop:        "; python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "multi-rank, native:     "; QADC_BENCH_FORCE_DIST=1 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "multi-rank, torch path: "; QADC_BENCH_FORCE_DIST=1 QADC_BENCH_NATIVE_DIST=0 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done
