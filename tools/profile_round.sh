#!/bin/bash
# Per-round evidence (GPU box, via gpurun; written in round 5, tag = first argument).  The headline of bench.py is the metric's mode
# (32-query steps, ONE QUERY PER PASS), so:
#   (1) `rocprofv3 --kernel-trace --stats` of the headline + batched regions only (side legs off): the average duration of
#       scan_i8_kernel<16,true,true,false> there is what `roofline.avg_launch_ms` of that run's line must agree with
#   (2) PMC passes of `bench.py --pmc-leg` (two steps of the headline's mode): FETCH_SIZE, WRITE_SIZE (separate passes), SQ / GRBM set
#   (3) the same for the batched mode (QADC_BENCH_BATCHED loop only; as tools/profile_r02.sh did)
#   (4) the default command under the kernel trace (every leg), the plain bench line
#   (5) the IVF legs alone: kernel trace + PMC passes (as tools/profile_r04.sh), the one-of-8 stand-ins of both modes
# Then: python3 tools/summarize_profile_round.py <tag>   (writes profiles/<tag>_*)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
# raw traces are hundreds of MB: they stay under /tmp on the box; only the summaries (gpurun_out/profiles_<tag>/) and the small logs travel back
RAW=/tmp/qadc_prof_raw
rm -rf $RAW; mkdir -p $RAW $R/gpurun_out
cd /tmp
OFF="QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_32X4=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_C2=0"
# (1)
env $OFF true
( export $OFF; rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof_${TAG}_headline_kt -- python3 $R/bench.py --steps 20 --warmup 2 > $RAW/prof_${TAG}_headline_kt.log 2>&1 )
# (2)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $RAW/prof_${TAG}_headline_$C -- python3 $R/bench.py --pmc-leg > $RAW/prof_${TAG}_headline_$C.log 2>&1
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $RAW/prof_${TAG}_headline_sq -- python3 $R/bench.py --pmc-leg > $RAW/prof_${TAG}_headline_sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $RAW/prof_${TAG}_headline_sq2 -- python3 $R/bench.py --pmc-leg > $RAW/prof_${TAG}_headline_sq2.log 2>&1
# (4)
QADC_BENCH_CPU_SECONDS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof_${TAG}_kt -- python3 $R/bench.py --steps 5 --warmup 1 > $RAW/prof_${TAG}_kt.log 2>&1
# (5)
for SH in c3 c5; do
  X=ivf; [ $SH = c5 ] && X=ivfc5
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof_${TAG}_${X}_kt -- python3 $R/tools/ivf_shard_one.py $SH none > $RAW/prof_${TAG}_${X}_kt.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $RAW/prof_${TAG}_${X}_$C -- python3 $R/tools/ivf_shard_one.py $SH none > $RAW/prof_${TAG}_${X}_$C.log 2>&1
  done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $RAW/prof_${TAG}_${X}_sq -- python3 $R/tools/ivf_shard_one.py $SH none > $RAW/prof_${TAG}_${X}_sq.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $RAW/prof_${TAG}_${X}_sq2 -- python3 $R/tools/ivf_shard_one.py $SH none > $RAW/prof_${TAG}_${X}_sq2.log 2>&1
done
cd $R
python3 bench.py > $RAW/${TAG}_bench_plain.json 2> $RAW/${TAG}_bench_plain.err
bash tools/dist_sizes.sh > $RAW/${TAG}_shard_sizes.txt 2>&1
RANKS_EMU=0 python3 tools/ivf_shard_sizes.py > $RAW/${TAG}_ivf_shard_sizes.txt 2>/dev/null
QADC_PROF_RAW=$RAW QADC_PROFILES_OUT=$R/gpurun_out/profiles_${TAG} python3 tools/summarize_profile_round.py $TAG > gpurun_out/${TAG}_summary.log 2>&1
tail -n 40 gpurun_out/${TAG}_summary.log
