# One of 8 ranks' flat step (125M-code shard of the 1B list, 32 queries) on ONE GPU:
#   single-GPU loop on the shard (no merge) | native merge standing in for rank 0 of 8 (loopback transport: the merge
#   replays 8 ranks' worth of streams) with the host-share replay | the same with the device merge
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0
P='import sys,json; j=json.loads(sys.stdin.read()); print("%.4f ms/step  %.3e codes/s  %s" % (j["ms_per_step"], j["value"], j.get("multi_gpu_merge")))'
for i in 1 2; do
echo -n "single-GPU loop, 125M codes:             "; QADC_BENCH_CODES=125e6 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "rank 0 of 8 (loopback), host share:      "; QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "rank 0 of 8 (loopback), device merge:    "; QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 QADC_BENCH_DIST_OPTS=dist_device_nq=1 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done
