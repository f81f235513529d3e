# One of 8 ranks' flat step on ONE GPU, BOTH scan modes of the 32-query step from the same bench.py run (round 5: the headline is the
# metric's mode, one query per pass; the batched mode is `value_batched`):
#   whole 1B list on one GPU | single-GPU loop on a 125M-code shard (no merge) | native merge standing in for rank 0 of 8 (loopback
#   transport: this process scans 1/8 of the list, pre-scans 1/8 of the starts and merges 8 ranks' worth of streams)
# Linear scaling would put the one-of-8 step at 1/8 of the 1B step.
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_C2=0
P='import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("one query per pass %.4f ms/step (%.3e codes/s, roofline.frac %.3f, kernel %.4f ms/step) | batched %.4f ms/step (%.3e codes/s)" % (j["ms_per_step"], j["value"], r["frac"], j["phases"]["scan_kernel_ms_per_step"], j["ms_per_step_batched"], j["value_batched"]))'
for i in 1 2; do
echo -n "1B list, one GPU:                    "; QADC_BENCH_CODES=1e9 python3 bench.py --steps 20 --warmup 3 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "single-GPU loop, 125M codes:         "; QADC_BENCH_CODES=125e6 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
echo -n "rank 0 of 8 (loopback), host share:  "; QADC_BENCH_CODES=1e9 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_LOOPBACK_WORLD=8 python3 bench.py --steps 60 --warmup 5 2>/dev/null | grep "^{" | python3 -c "$P"
done
