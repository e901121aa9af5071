#!/bin/bash
# rocprofv3 kernel trace of the headline solve (n = 371): span, main-queue busy time and kernel count of EVERY time step of the last solve,
# the per-class totals of three step ranges, and the kernels of one step (argument, default 5).   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-path > /dev/null 2> gpurun_out/prof_t.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" ${1:-5} <<'PY'
import csv,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1]))); which=int(sys.argv[2])
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_dense_residual" in r["Kernel_Name"]][-45:]
S=lambda r:int(r["Start_Timestamp"]); E=lambda r:int(r["End_Timestamp"])
nm=lambda r:r["Kernel_Name"].split('(')[0].replace('void ','').replace('dre::','')[:40]
# the solve starts well before the first residual kernel (factorisations): find the idle gap in front of it
first=idx[0]
j=first
while j>0 and S(rows[j])-E(rows[j-1])<2_000_000: j-=1
bounds=[S(rows[j])]+[S(rows[i]) for i in idx[1:]]
last=idx[-1]; k=last
while k+1<len(rows) and S(rows[k+1])-E(rows[k])<2_000_000: k+=1
bounds.append(E(rows[k]))
busyq=defaultdict(float)
for r in rows[j:k+1]: busyq[r["Queue_Id"]]+=E(r)-S(r)
mainq=max(busyq,key=busyq.get)
print("solve span ms",(bounds[-1]-bounds[0])/1e6,"busy per queue ms",{q:round(v/1e6,2) for q,v in busyq.items()})
per=[]
for s in range(45):
    sel=[r for r in rows[j:k+1] if bounds[s]<=S(r)<bounds[s+1]]
    mb=sum(E(r)-S(r) for r in sel if r["Queue_Id"]==mainq)
    per.append(((bounds[s+1]-bounds[s])/1e3, mb/1e3, len(sel), sum(1 for r in sel if r["Queue_Id"]==mainq)))
    print(f"step {s+1:2d}: span {per[-1][0]:7.1f} us  main busy {per[-1][1]:7.1f}  kernels {per[-1][2]:4d} (main {per[-1][3]})")
for lo,hi in ((1,1),(2,12),(13,45)):
    c=defaultdict(lambda:[0,0.0])
    for r in rows[j:k+1]:
        if bounds[lo-1]<=S(r)<bounds[hi] and r["Queue_Id"]==mainq: c[nm(r)][0]+=1; c[nm(r)][1]+=(E(r)-S(r))/1e3
    print(f"--- steps {lo}..{hi}: span {(bounds[hi]-bounds[lo-1])/1e3:.0f} us, main-queue classes")
    for kk,v in sorted(c.items(), key=lambda t:-t[1][1])[:16]: print(f"   {kk:42s} {v[0]:4d} {v[1]:8.0f} {v[1]/v[0]:6.1f}")
print(f"--- kernels of step {which}")
prev=None
for r in rows[j:k+1]:
    if not (bounds[which-1]<=S(r)<bounds[which]): continue
    gap=""
    if r["Queue_Id"]==mainq:
        if prev is not None and S(r)-prev>8000: gap=f"   <== main idle {(S(r)-prev)/1e3:.1f}"
        prev=E(r)
    print(f"{(S(r)-bounds[which-1])/1e3:8.1f} +{(E(r)-S(r))/1e3:6.1f} q{r['Queue_Id']} {nm(r)}{gap}")
PY
rm -rf gpurun_out/prof_t
