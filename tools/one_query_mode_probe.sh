set -x
export QADC_BENCH_CPU_SECONDS=0 QADC_BENCH_REAL_CODES=0 QADC_BENCH_IVF_CODES=0 QADC_BENCH_32X4=0 QADC_BENCH_LATENCY=0 QADC_BENCH_PMC=0 QADC_BENCH_C2=0 QADC_BENCH_SINGLE_QUERIES=16
QADC_BENCH_OPTS=share_variant=0,mq=0,front_run_max=0 python bench.py --steps 10 --warmup 2 > gpurun_out/probe_oneq.json 2> gpurun_out/probe_oneq.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/probe_oneq.json'))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "per query", d["ms_per_step"]/32)
print("batched roofline obj:", json.dumps(d["roofline_batched"])[:600])
print("single:", d["value_one_query_per_pass"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"])
print(d["phases"])
PY
cd /tmp && export TMPDIR=/tmp
QADC_BENCH_SINGLE_QUERIES=0 QADC_BENCH_OPTS=share_variant=0,mq=0,front_run_max=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcA -- python3 /root/repo/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pmcA/**/*counter_collection.csv',recursive=True)[0]
per={}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE":
        k=r["Kernel_Name"][:60]
        per.setdefault(k,[0,set()]); per[k][0]+=float(r["Counter_Value"]); per[k][1].add(r["Dispatch_Id"])
for k,(v,ids) in sorted(per.items(), key=lambda x:-x[1][0])[:6]:
    print(k, "dispatches",len(ids),"GB (x2 corrected)", v*1024*2/1e9)
PY
