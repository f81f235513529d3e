cd /tmp; export TMPDIR=/tmp
for w in 256 64 32; do
  rm -rf /tmp/pw_$w
  QADC_BENCH_OPTS=wgs_per_item=$w rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pw_$w -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-leg > /tmp/pw_$w.log 2>&1
  python3 - $w <<'PY'
import csv,glob,json,sys
w=sys.argv[1]
f=glob.glob('/tmp/pw_%s/**/*counter_collection.csv'%w,recursive=True)[0]
tot=0;n=set();dur=0
for r in csv.DictReader(open(f)):
    if "scan_i8_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
        tot+=float(r["Counter_Value"]); 
        if r["Dispatch_Id"] not in n: dur+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        n.add(r["Dispatch_Id"])
leg=[json.loads(l) for l in open('/tmp/pw_%s.log'%w) if l.startswith("{") and "pmc_leg" in l][-1]
alg=leg["scan_codes"]*8
print("wgs_per_item=%s: launches %d, FETCH x2048 = %.1f GB, algorithmic %.1f GB, ratio %.3f, kernel time %.2f ms -> HBM read %.0f GB/s, algorithmic %.0f GB/s" % (w,len(n),tot*2048/1e9,alg/1e9,tot*2048/alg,dur/1e6,tot*2048/dur,alg/dur))
PY
done
