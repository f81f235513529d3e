# which earlier leg of bench.py makes the 1024-query IVF leg slow?
P='import sys,json; j=json.loads(sys.stdin.read()); print("ivf %.3f / %.3f" % (j["ivf"]["us_per_query"], j["ivf"]["us_per_query_at_2048_query_batches"]))'
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_C5=0 QADC_BENCH_LATENCY=0
echo -n "all legs:            "; python3 bench.py 2>/dev/null | grep '^{' | python3 -c "$P"
echo -n "no PMC children:     "; QADC_BENCH_PMC=0 python3 bench.py 2>/dev/null | grep '^{' | python3 -c "$P"
echo -n "no 32x4:             "; QADC_BENCH_PMC=0 QADC_BENCH_32X4=0 python3 bench.py 2>/dev/null | grep '^{' | python3 -c "$P"
echo -n "no single queries:   "; QADC_BENCH_PMC=0 QADC_BENCH_32X4=0 QADC_BENCH_SINGLE_QUERIES=0 python3 bench.py 2>/dev/null | grep '^{' | python3 -c "$P"
echo -n "small headline list: "; QADC_BENCH_PMC=0 QADC_BENCH_32X4=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_CODES=2e7 python3 bench.py 2>/dev/null | grep '^{' | python3 -c "$P"
