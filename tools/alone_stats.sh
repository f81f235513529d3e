# Kernel stats of bench.ivf_leg with ONE batch in flight (every kernel nearly alone on the GPU): tools/alone_stats.sh c3|c5  -> gpurun_out/alone_<shape>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
QADC_BENCH_IVF_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/alone_$1 -- python3 $R/tools/ivf_shard_one.py $1 none > $R/gpurun_out/alone_$1.log 2>&1
python3 - $R/gpurun_out/alone_$1 > $R/gpurun_out/alone_$1.txt <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "").replace("qadc::", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
    print("%-52s calls %6s avg %9.1f us  min %9.1f  %6s %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $R/gpurun_out/alone_$1
cat $R/gpurun_out/alone_$1.txt
