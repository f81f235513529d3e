# round 5: candidate layouts with FOUR highest-priority streams only (IVF scan stream W, front, collectives, merge), copy and / or
# ordering demoted to normal priority — the legs that decided against candidates 1 and 2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
OFF="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0"
PF='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("one query per pass %.3f ms/step | batched %.3f ms/step | %s" % (j["ms_per_step"], j["ms_per_step_batched"], j["stream_layout"]))'
PI='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
legs() {
  echo -n "[$1] flat 125M, one GPU: "; env $OFF QADC_BENCH_CODES=125e6 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | python3 -c "$PF"
  echo -n "[$1] flat, rank 0 of 8 (loopback): "; env $OFF QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | python3 -c "$PF"
  for shape in c3 c5; do
    echo -n "[$1] IVF $shape one GPU: "; python3 $R/tools/ivf_shard_one.py $shape none 2>/dev/null | python3 -c "$PI"
    for hist in none torch_before; do
      echo -n "[$1] IVF $shape one of 8 ranks, history=$hist: "; QADC_PROBE_RCCL=$hist python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$PI"
    done
  done
}
export QADC_W_PRIO=high QADC_MERGE_PRIO2=high QADC_WGQ_STREAM=1
for order in W,C,O,F,S,L,M0 S,C,O,F,W,L,M0; do
  export QADC_SHARED_STREAM_ORDER=$order
  QADC_C_PRIO=normal QADC_O_PRIO=normal legs "$order, W M high, C O normal"
  QADC_O_PRIO=normal legs "$order, W M high, O normal"
  QADC_C_PRIO=normal legs "$order, W M high, C normal"
done
