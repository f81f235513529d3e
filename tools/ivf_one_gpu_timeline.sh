cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlc3
rocprofv3 --kernel-trace --output-format csv -d /tmp/tlc3 -- python3 $GRAFT_REPO_ROOT/tools/ivf_shard_one.py c3 none > /tmp/tlc3.log 2>&1; true
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/tlc3/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),int(r["Queue_Id"]),r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("qadc::","")[:40]) for r in rows]
ev.sort()
# steady-state window: the 1024-query batches region: take head launches (scan_query_kernel with HEAD) on the scan queue
heads=[e for e in ev if e[3].startswith("scan_query_kernel")]
# pick window from 20th to 60th head
t0=heads[20][0]; t1=heads[60][0]
print("window %.3f ms, %d batches -> %.3f ms/batch" % ((t1-t0)/1e6, 40, (t1-t0)/1e6/40))
byq=collections.defaultdict(list)
for s,e,q,n in ev:
    if e>t0 and s<t1: byq[q].append((max(s,t0),min(e,t1),n))
for q,l in sorted(byq.items()):
    busy=0; last=t0
    l.sort()
    # union length
    cur_s,cur_e=None,None; tot=0
    for s,e,n in l:
        if cur_e is None or s>cur_e:
            if cur_e is not None: tot+=cur_e-cur_s
            cur_s,cur_e=s,e
        else: cur_e=max(cur_e,e)
    if cur_e is not None: tot+=cur_e-cur_s
    names=collections.Counter(n for _,_,n in l)
    per=collections.defaultdict(float)
    for s,e,n in l: per[n]+=(e-s)/1e6/40
    print("queue %d: busy %.1f%% | per batch ms: %s" % (q, 100*tot/(t1-t0), ", ".join("%s %.3f" % (k,v) for k,v in sorted(per.items(), key=lambda x:-x[1])[:6])))
# gaps on the scan queue between consecutive kernels
sq=max(byq, key=lambda q: sum(1 for x in byq[q] if x[2].startswith("scan_i8_mq")))
l=sorted(byq[sq]); gaps=[]
for (s0,e0,n0),(s1,e1,n1) in zip(l,l[1:]):
    if s1>e0: gaps.append(((s1-e0)/1e3,n0,n1))
import statistics
tot=sum(g[0] for g in gaps)
print("scan queue %d: %d gaps, total %.3f ms per batch" % (sq,len(gaps),tot/1e3/40))
agg=collections.defaultdict(list)
for g,n0,n1 in gaps: agg[(n0,n1)].append(g)
for k,v in sorted(agg.items(), key=lambda x:-sum(x[1]))[:8]:
    print("   %-40s -> %-40s n=%d mean %.1f us total/batch %.1f us" % (k[0],k[1],len(v),sum(v)/len(v),sum(v)/40))
PY
