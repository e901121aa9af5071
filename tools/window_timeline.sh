#!/bin/bash
# rocprofv3 kernel trace of bench.py: all kernels (every queue) in a window of the LAST solve.  usage: tools/window_timeline.sh <frac> <len_us> [bench args]
# frac = position of the window start inside the last solve (0..1)        (run on the GPU box)
frac=$1; len=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
if [ -n "$WT_SCRIPT" ]; then      # another driver script instead of bench.py (e.g. WT_SCRIPT=tools/time_default_adi.py with its own arguments)
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python $WT_SCRIPT "$@" > /dev/null 2> gpurun_out/prof_t.err
else
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> gpurun_out/prof_t.err
fi
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" $frac $len <<'PY'
import csv,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1]))); frac=float(sys.argv[2]); ln=float(sys.argv[3])*1e3
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# solves are separated by long idle gaps (host work between solves): find the last gap > 2 ms
starts=[int(r["Start_Timestamp"]) for r in rows]; ends=[int(r["End_Timestamp"]) for r in rows]
segs=[[rows[0]]]; hi=ends[0]
for i in range(1,len(rows)):
    if starts[i]-hi>1_000_000: segs.append([])
    segs[-1].append(rows[i]); hi=max(hi,ends[i])
big=[g for g in segs if len(g)>1000]
seg=big[-1] if big else max(segs,key=len)
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
print("last solve span ms", (t1-t0)/1e6, "kernels", len(seg))
busy=defaultdict(float)
for r in seg: busy[r["Queue_Id"]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
print("busy ms per queue", {k:round(v,2) for k,v in busy.items()})
import os
if os.environ.get("WT_QUEUE_KERNELS"):
    cnt=defaultdict(lambda: defaultdict(lambda:[0,0.0]))
    for r in seg:
        c=cnt[r["Queue_Id"]][r["Kernel_Name"].split('(')[0].replace('void ','')[:44]]; c[0]+=1; c[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for q,d in cnt.items():
        print("queue",q, sorted(((k,v[0],round(v[1])) for k,v in d.items()), key=lambda t:-t[2])[:6])
w0=t0+frac*(t1-t0); w1=w0+ln
for r in seg:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s<w0 or s>w1: continue
    print(f"{(s-w0)/1e3:9.1f} +{(e-s)/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'].split('(')[0].replace('void ','')[:50]} g{r.get('Grid_Size','')}")
PY
rm -rf gpurun_out/prof_t
