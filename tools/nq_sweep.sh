# Queries per step: the 1B-code step on one GPU and one of 8 ranks' step (loopback merge), NQ = 32 / 64 / 128
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s" % (j["ms_per_step"], j["value"]))'
for nq in 32 64 128; do
echo -n "NQ=$nq one GPU, 1B codes:        "; QADC_BENCH_NQ=$nq python3 bench.py --steps 20 --warmup 3 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "NQ=$nq one of 8 ranks (loopback): "; QADC_BENCH_NQ=$nq QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 bench.py --steps 40 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done
