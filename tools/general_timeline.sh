#!/bin/bash
# rocprofv3 kernel trace of one general-path solve (tools/trace_one.py n steps): per-kernel summary + a window of the steady state.
# usage: tools/general_timeline.sh <n> <steps> <frac> <len_us>          (run on the GPU box)
n=$1; steps=$2; frac=$3; len=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python tools/trace_one.py $n $steps > gpurun_out/trace_one_$n.log 2> gpurun_out/prof_t.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" $frac $len <<'PY'
import csv,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1]))); frac=float(sys.argv[2]); ln=float(sys.argv[3])*1e3
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
starts=[int(r["Start_Timestamp"]) for r in rows]; ends=[int(r["End_Timestamp"]) for r in rows]
segs=[[rows[0]]]; hi=ends[0]
for i in range(1,len(rows)):
    if starts[i]-hi>20_000_000: segs.append([])
    segs[-1].append(rows[i]); hi=max(hi,ends[i])
big=[g for g in segs if len(g)>1000]
seg=big[-1] if big else max(segs,key=len)
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
print("last solve span ms", (t1-t0)/1e6, "kernels", len(seg))
busy=defaultdict(float)
for r in seg: busy[r["Queue_Id"]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
print("busy ms per queue", {k:round(v,2) for k,v in busy.items()})
cnt=defaultdict(lambda:[0,0.0])
for r in seg:
    c=cnt[r["Kernel_Name"].split('(')[0].replace('void ','')[:48]]; c[0]+=1; c[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print("| kernel | calls | total us | avg us |")
for k,v in sorted(cnt.items(), key=lambda t:-t[1][1])[:45]:
    print(f"| {k} | {v[0]} | {v[1]:.0f} | {v[1]/v[0]:.1f} |")
import os
with open(os.environ.get("TL_DUMP","gpurun_out/tl_last_solve.csv"),"w") as fo:
    for r in seg:
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        fo.write(f"{(s-t0)/1e3:.1f},{(e-s)/1e3:.1f},{r['Queue_Id']},{r['Kernel_Name'].split('(')[0].replace('void ','').replace('dre::','')[:40]},{r.get('Grid_Size','')},{r.get('Workgroup_Size','')}\n")
w0=t0+frac*(t1-t0); w1=w0+ln
for r in seg:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s<w0 or s>w1: continue
    print(f"{(s-w0)/1e3:9.1f} +{(e-s)/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'].split('(')[0].replace('void ','')[:50]} g{r.get('Grid_Size','')}")
PY
rm -rf gpurun_out/prof_t
