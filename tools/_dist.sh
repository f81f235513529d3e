export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
run() { python bench.py --steps $1 --warmup 10 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s ms/step %.3f value %.3e frac %.3f prescan %.3f recall %.2f' % ('$2', d['ms_per_step'], d['value'], d['roofline']['frac'], d['phases']['prescan_quantize_ms_per_step'], d['recall_at_100']))"; }
for rep in 1 2; do
QADC_BENCH_CODES=125e6 run 200 "125M"
QADC_BENCH_CODES=125e6 QADC_BENCH_FORCE_DIST=1 run 200 "125M dist(world 1)"
done
QADC_BENCH_FORCE_DIST=1 run 30 "1B dist(world 1)"
run 30 "1B"
