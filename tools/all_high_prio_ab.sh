# round 5: every stream of the set at the HIGHEST priority (scan and merge too) against the default (scan, merge lowest), same box:
# flat 1B headline + batched, one of 8 ranks' flat step, IVF legs on one GPU, one of 8 ranks' IVF batch in two process histories.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
OFF="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0"
PF='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("one query per pass %.3f ms/step (roofline %.3f) | batched %.3f ms/step" % (j["ms_per_step"], j["roofline"]["frac"], j["ms_per_step_batched"]))'
PI='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
for rep in 1 2; do
for prio in "default" "high"; do
  if [ $prio = high ]; then export QADC_SCAN_PRIO=high QADC_MERGE_PRIO2=high; else unset QADC_SCAN_PRIO QADC_MERGE_PRIO2; fi
  echo -n "[$prio] flat 1B, one GPU: "; env $OFF python3 $R/bench.py --steps 20 --warmup 3 2>/dev/null | python3 -c "$PF"
  echo -n "[$prio] flat, rank 0 of 8 (loopback): "; env $OFF QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | python3 -c "$PF"
  for shape in c3 c5; do
    echo -n "[$prio] IVF $shape one GPU: "; python3 $R/tools/ivf_shard_one.py $shape none 2>/dev/null | python3 -c "$PI"
    for hist in none torch_before; do
      echo -n "[$prio] IVF $shape one of 8 ranks, history=$hist: "; QADC_PROBE_RCCL=$hist python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$PI"
    done
  done
done
done
