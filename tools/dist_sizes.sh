# bench.py at the shard sizes of 1/2/4/8 ranks, plain and through the multi-rank code path with world 1 (RCCL)
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
run() { python bench.py --steps $1 --warmup 5 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s ms/step %.3f value %.3e lds %.3f scan_ms %.3f prescan %.3f' % ('$2', d['ms_per_step'], d['value'], d['roofline']['lds']['frac'], d['phases']['scan_kernel_ms_per_step'], d['phases']['prescan_quantize_ms_per_step']))"; }
for n in 1e9 5e8 2.5e8 1.25e8; do
QADC_BENCH_CODES=$n run 40 "$n"
QADC_BENCH_CODES=$n QADC_BENCH_FORCE_DIST=1 run 40 "$n dist(world 1)"
done
