R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.txt 2>&1; grep -a "passed\|failed\|rror" gpurun_out/r05_gpu_suite.txt | tail -3
PI='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048  %s" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"], j["stream_layout"]))'
for shape in c3 c5; do for hist in none torch_before; do echo -n "IVF $shape one of 8 ranks, history=$hist: "; QADC_PROBE_RCCL=$hist python3 $R/tools/ivf_shard_one.py $shape range 0 2>/dev/null | python3 -c "$PI"; done; done
bash tools/dist_sizes_r05.sh 2>&1 | head -3
