# IVF leg only, with option sets given as arguments (QADC_BENCH_IVF_OPTS syntax)
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_CODES=2e7
for rep in 1 2; do for o in "$@"; do
echo -n "[$o] "; QADC_BENCH_IVF_OPTS="$o" python3 bench.py --steps 2 --warmup 1 2>/dev/null | grep '^{' | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())['ivf']
print('us/query %.3f  at 2048: %.3f  grouped %d fallbacks %d' % (j['us_per_query'], j['us_per_query_at_2048_query_batches'], j['batches_through_partition_major_second_phase'], j['of_them_redone_on_the_level_path']))"
done; done
