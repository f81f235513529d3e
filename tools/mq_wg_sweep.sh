# Partition-major second phase against codes per workgroup (option mq_codes_per_wg): tools/ivf_head_cycles.py, one batch at a time.
# -> gpurun_out/mq_wg_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mq_wg_sweep.txt
: > $OUT
for rep in 1 2; do
for v in 4096 8192 12288 16384 32768 65536; do
  for shape in c3 c5; do
    echo -n "mq_codes_per_wg $v " >> $OUT
    timeout 300 python3 $R/tools/ivf_head_cycles.py $shape mq_codes_per_wg=$v 2>&1 | tail -1 >> $OUT
  done
done
done
cat $OUT
