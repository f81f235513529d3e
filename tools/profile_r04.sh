#!/bin/bash
# Round-4 evidence (GPU box, via gpurun; round 3's script with a tag and BOTH IVF shapes): everything tools/profile_r02.sh collects, plus
#   - a kernel trace of ONLY the headline loop (roofline_batched.avg_launch_ms must be recomputable from it),
#   - a kernel trace and PMC passes of ONLY the IVF leg at the BASELINE configs[2] shape (scan_query_kernel head,
#     scan_i8_mq_kernel grouped phase: FETCH_SIZE / WRITE_SIZE in their own passes, then the SQ / GRBM set),
#   - the plain bench line, the one-of-8-rank step times (flat: tools/dist_sizes3.sh, IVF: tools/ivf_shard_sizes.py).
# Then: python3 tools/summarize_profile_r04.py   (copies the summaries into profiles/)
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
bash $R/tools/profile_r02.sh $TAG > $R/gpurun_out/profile_${TAG}_base.log 2>&1
cd /tmp
(
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_32X4=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0
# (front_run_max = 0: every multi-query launch stays on the main stream inside the event-timed groups, so the launches the
#  bench line averages ARE the launches the kernel-stats CSV lists)
QADC_BENCH_OPTS=front_run_max=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_batched_kt -- python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_batched_kt.log 2>&1
)
for SH in c3 c5; do
  X=ivf; [ $SH = c5 ] && X=ivfc5
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${X}_kt -- python3 $R/tools/ivf_shard_one.py $SH none > $R/gpurun_out/prof_${TAG}_${X}_kt.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_${TAG}_${X}_$C -- python3 $R/tools/ivf_shard_one.py $SH none > $R/gpurun_out/prof_${TAG}_${X}_$C.log 2>&1
  done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_${X}_sq -- python3 $R/tools/ivf_shard_one.py $SH none > $R/gpurun_out/prof_${TAG}_${X}_sq.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_${X}_sq2 -- python3 $R/tools/ivf_shard_one.py $SH none > $R/gpurun_out/prof_${TAG}_${X}_sq2.log 2>&1
done
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_plain.json 2> gpurun_out/${TAG}_bench_plain.err
bash tools/dist_sizes3.sh > gpurun_out/${TAG}_shard_sizes.txt 2>&1
bash tools/stream_order_ab3.sh > /dev/null 2>&1; cp gpurun_out/stream_order3.txt gpurun_out/${TAG}_ivf_shard_sizes.txt
tail -n 3 gpurun_out/${TAG}_shard_sizes.txt gpurun_out/${TAG}_ivf_shard_sizes.txt
