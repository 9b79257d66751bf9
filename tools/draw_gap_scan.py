"""One draw of a sampler in a rocprofv3 --kernel-trace CSV (the kernels between the last-but-two and last-but-one launch of a marker
kernel): per-kernel time, busy time and gaps of the main queue, and what the side-stream kernels cost the ones they overlap.
usage: draw_gap_scan.py <dir or kernel_trace.csv> [marker kernel name = k_mh_accept]"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
# last draw: between the last two k_mh_accept
marker = sys.argv[2] if len(sys.argv) > 2 else "k_mh_accept"
acc=[i for i,r in enumerate(rows) if marker in r["Kernel_Name"]]
a,b=acc[-3],acc[-2]
seg=rows[a+1:b+1]
t0=int(seg[0]["Start_Timestamp"])
span=(int(seg[-1]["End_Timestamp"])-t0)/1e3
busy_main=0; prev_end=None; gaps=[]; 
from collections import defaultdict
per=defaultdict(lambda:[0,0.0])
for r in seg:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    per[short(r["Kernel_Name"])][0]+=1; per[short(r["Kernel_Name"])][1]+=(e-s)/1e3
qs=defaultdict(list)
for r in seg: qs[r.get("Queue_Id","?")].append(r)
print("draw span us", span, "kernels", len(seg), "queues", {q:len(v) for q,v in qs.items()})
for k,(n,t) in sorted(per.items(), key=lambda kv:-kv[1][1]): print(f"  {k:42s} {n:4d} {t:9.1f} us  avg {t/n:7.1f}")
# gaps on the main queue (the one with most kernels)
mq=max(qs,key=lambda q:len(qs[q])); m=qs[mq]
g=0
for x,y in zip(m[:-1],m[1:]):
    d=(int(y["Start_Timestamp"])-int(x["End_Timestamp"]))/1e3
    if d>0: g+=d
print("main queue busy", sum((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in m), "gaps", g)
# kd kernels overlapped by side-queue kernels vs not
side=[r for q,v in qs.items() if q!=mq for r in v]
def overl(r):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    return any(int(x["Start_Timestamp"])<e and int(x["End_Timestamp"])>s for x in side)
for name in ("k_kick_drift_v2","k_gauss_grad_v2"):
    o=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in m if name in r["Kernel_Name"] and overl(r)]
    n=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in m if name in r["Kernel_Name"] and not overl(r)]
    print(name, "overlapped", len(o), sum(o)/max(1,len(o)), "alone", len(n), sum(n)/max(1,len(n)))
for r in side:
    print("  side:", short(r["Kernel_Name"]), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
