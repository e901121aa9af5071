#!/bin/bash
# rocprofv3 kernel trace of configs[2] (tools/time_proj1357.py 1): per-kernel table of the last solve and a window of its timeline (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_p
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p -- python tools/time_proj1357.py 1 > gpurun_out/proj1357_run.log 2> gpurun_out/prof_p.err
f=$(find gpurun_out/prof_p -name "*kernel_trace.csv" | head -1)
python - "$f" ${1:-0.5} ${2:-1500} <<'PY'
import csv,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1]))); frac=float(sys.argv[2]); ln=float(sys.argv[3])*1e3
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
S=lambda r:int(r["Start_Timestamp"]); E=lambda r:int(r["End_Timestamp"])
segs=[[rows[0]]]; hi=E(rows[0])
for r in rows[1:]:
    if S(r)-hi>150_000_000: segs.append([])
    segs[-1].append(r); hi=max(hi,E(r))
big=[g for g in segs if len(g)>1000]
seg=big[-1] if big else max(segs,key=len)
t0=S(seg[0]); t1=max(E(r) for r in seg)
print("last solve span ms",(t1-t0)/1e6,"kernels",len(seg))
busy=defaultdict(float)
for r in seg: busy[r["Queue_Id"]]+=(E(r)-S(r))/1e6
print("busy ms per queue",{k:round(v,2) for k,v in busy.items()})
cnt=defaultdict(lambda:[0,0.0])
nm=lambda r:r["Kernel_Name"].split('(')[0].replace('void ','').replace('dre::','')[:48]
for r in seg: c=cnt[nm(r)]; c[0]+=1; c[1]+=(E(r)-S(r))/1e3
for k,v in sorted(cnt.items(), key=lambda t:-t[1][1])[:50]: print(f"| {k} | {v[0]} | {v[1]:.0f} | {v[1]/v[0]:.1f} |")
w0=t0+frac*(t1-t0); w1=w0+ln
for r in seg:
    if S(r)<w0 or S(r)>w1: continue
    print(f"{(S(r)-w0)/1e3:9.1f} +{(E(r)-S(r))/1e3:7.1f} q{r['Queue_Id']} {nm(r)} g{r.get('Grid_Size','')}")
PY
rm -rf gpurun_out/prof_p
