# multi-rank loop (one rank): where are the slow iterations at different step / warm-up counts?
export QADC_BENCH_CODES=125e6 QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_FORCE_DIST=1 QADC_BENCH_STEP_LOG=gpurun_out/steplog.txt
rm -f gpurun_out/steplog.txt
for sw in "100 5" "100 5" "20 3" "20 3" "50 10"; do set -- $sw
echo "[steps $1 warmup $2]" >> gpurun_out/steplog.txt
python3 bench.py --steps $1 --warmup $2 2>gpurun_out/stall_err.txt | grep "^{" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("mean %.4f ms/step" % (j["ms_per_step"]))' >> gpurun_out/steplog.txt
grep SLOW gpurun_out/stall_err.txt >> gpurun_out/steplog.txt
done
cat gpurun_out/steplog.txt
