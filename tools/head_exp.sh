# IVF head experiments (round 4): the head workgroup's phase clocks and the three launches of a grouped batch, per variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/head_exp.txt
: > $OUT
for shape in c3 c5; do
  for opts in "" "wgq_variant=1" "wgq_variant=2" "wgq_variant=3" $EXTRA_OPTS; do
    echo -n "[$opts] " >> $OUT
    timeout 300 python3 $R/tools/ivf_head_cycles.py $shape $opts 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
