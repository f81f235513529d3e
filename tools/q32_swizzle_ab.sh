# Round 5: the 16-replica 32x4 image read WITHOUT its 2-way bank conflict (QADC_Q32_SWIZZLE, qadc_query_kernel.hip): parity of the
# 32x4 paths, same-box A/B against a build with -DQADC_Q32_SWIZZLE=0 (make ab AB_QUERY_FLAGS=-DQADC_Q32_SWIZZLE=0 AB_KERNEL_FLAGS=),
# and the LDS counters of the C5 head under both.   -> gpurun_out/q32_swizzle_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/q32_swizzle_ab.txt
: > $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python3 -m pytest tests/test_gpu_config_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -n 4 -m gpu -k "32 or c5 or C5 or fuzz" 2>&1 | grep -E "passed|failed|error" | tail -3 >> $OUT
P='import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("%.3f ms/batch  %.3f us/q@2048" % (j["ms_per_batch"], j["us_per_query_at_2048_query_batches"]))'
export QADC_TEST_HOOKS=1
for rep in 1 2; do
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  echo -n "$lib c5 head cycles: " >> $OUT
  QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_head_cycles.py c5 2>&1 | tail -1 >> $OUT
  for place in none range; do
    echo -n "$lib c5 $place: " >> $OUT
    QADC_LIB_PATH=$R/quick-adc_amd/$lib timeout 300 python3 $R/tools/ivf_shard_one.py c5 $place 0 2>/dev/null | python3 -c "$P" >> $OUT 2>&1
  done
done
done
cd /tmp
for lib in libqadc_hip.so libqadc_hip_nopipe.so; do
  export QADC_LIB_PATH=$R/quick-adc_amd/$lib
  rm -rf /tmp/swz_$lib
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/swz_$lib -- python3 $R/tools/ivf_shard_one.py c5 none > /tmp/swz_$lib.log 2>&1
  rm -rf /tmp/swzkt_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/swzkt_$lib -- python3 $R/tools/ivf_shard_one.py c5 none > /tmp/swzkt_$lib.log 2>&1
  python3 - $lib >> $OUT <<'PY'
import sys, glob, csv, collections
lib = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('/tmp/swz_%s/**/*counter_collection.csv' % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void qadc::', '').split('(')[0][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
for k, c in acc.items():
    if 'scan_query' in k or 'mq_narrow' in k:
        print("%s %-60s launches %d  lds_conflict_fraction %.3f  lds_busy %.3f" % (lib, k, n[k], c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), c['SQ_LDS_IDX_ACTIVE'] / max(c['SQ_BUSY_CYCLES'], 1)))
for f in glob.glob('/tmp/swzkt_%s/**/*kernel_stats.csv' % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'scan_query' in r['Name'] or 'mq_narrow' in r['Name']:
            print("%s %-60s calls %s avg %.1f us" % (lib, r['Name'].replace('void qadc::(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $OUT
