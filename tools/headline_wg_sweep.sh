# round 5: the headline's mode (32-query steps, one query per pass) against workgroups per (query, level) item (option wgs_per_item;
# default: min(1024, 8192 / queries) = 256 at 32 queries per step).  -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0 QADC_BENCH_BATCHED=0
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/step  %.4e codes/s  roofline %.4f" % (j["ms_per_step"], j["value"], j["roofline"]["frac"]))'
for rep in 1 2; do
for w in 0 64 128 192 256 384 512 1024; do
  echo -n "wgs_per_item=$w: "
  if [ $w = 0 ]; then python3 $R/bench.py --steps 12 --warmup 2 2>/dev/null | python3 -c "$P"; else QADC_BENCH_OPTS=wgs_per_item=$w python3 $R/bench.py --steps 12 --warmup 2 2>/dev/null | python3 -c "$P"; fi
done
done
