# round 5: the whole bench line (every leg) under the candidate stream priorities against the defaults, two repetitions each
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp QADC_TEST_HOOKS=1 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_PMC=0
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("headline %.3f ms (%.3f) | batched %.3f | per-call %.3f | 32x4 %.3f | ivf %.3f / %.3f us | c5 %.3f / %.3f us | c2 %.3f ms, %.4f ms | latency %.1f us | %s" % (j["ms_per_step"], j["roofline"]["frac"], j["ms_per_step_batched"], j["roofline_one_query_per_call"]["frac"], j["roofline_32x4"]["frac"], j["ivf"]["us_per_query"], j["ivf"]["us_per_query_at_2048_query_batches"], j["ivf_c5_one_gpu"]["us_per_query"], j["ivf_c5_one_gpu"]["us_per_query_at_2048_query_batches"], j["c2"]["batched"]["ms_per_step"], j["c2"]["one_query_per_pass"]["ms_per_query"], j["latency_us_single_query"]["value"], j["stream_layout"]))'
for rep in 1 2; do
  unset QADC_W_PRIO QADC_MERGE_PRIO2 QADC_WGQ_STREAM QADC_C_PRIO QADC_O_PRIO
  echo -n "[default]   "; python3 $R/bench.py 2>/dev/null | python3 -c "$P"
  export QADC_W_PRIO=high QADC_MERGE_PRIO2=high QADC_WGQ_STREAM=1 QADC_C_PRIO=normal QADC_O_PRIO=normal
  echo -n "[candidate] "; python3 $R/bench.py 2>/dev/null | python3 -c "$P"
done
