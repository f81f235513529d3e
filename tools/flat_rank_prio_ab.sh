R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1
OFF="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0"
PF='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("one query per pass %.3f ms/step | batched %.3f ms/step | %s" % (j["ms_per_step"], j["ms_per_step_batched"], j["stream_layout"]))'
run() { echo -n "[$1] flat, rank 0 of 8 (loopback): "; env $OFF QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 $R/bench.py --steps 60 --warmup 5 2>/dev/null | python3 -c "$PF"; }
for rep in 1 2; do
unset QADC_SHARED_STREAM_ORDER QADC_W_PRIO QADC_MERGE_PRIO2; run "default"
QADC_MERGE_PRIO2=high run "M high"
QADC_W_PRIO=high run "W high"
QADC_W_PRIO=high QADC_MERGE_PRIO2=high run "W high, M high"
QADC_SHARED_STREAM_ORDER=W,C,O,F,S,L,M0 run "order W,C,O,F,S,L,M0"
QADC_SHARED_STREAM_ORDER=W,C,O,F,S,L,M0 QADC_W_PRIO=high run "order W,C,O,F,S,L,M0, W high"
QADC_SHARED_STREAM_ORDER=W,C,O,F,S,L,M0 QADC_W_PRIO=high QADC_MERGE_PRIO2=normal run "order W,C,O,F,S,L,M0, W high, M normal"
done
