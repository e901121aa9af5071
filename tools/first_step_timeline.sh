#!/bin/bash
# rocprofv3 kernel trace of bench.py (n = 371): the kernels of the FIRST time step of the last solve, with gaps (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-path > /dev/null 2> gpurun_out/prof_t.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# solves start with k_assemble bursts after a long gap; find last big gap (> 2 ms) = start of last (profiled) solve
starts=[int(r["Start_Timestamp"]) for r in rows]; ends=[int(r["End_Timestamp"]) for r in rows]
segs=[[rows[0]]]; hi=ends[0]
for i in range(1,len(rows)):
    if starts[i]-hi>1_000_000: segs.append([])
    segs[-1].append(rows[i]); hi=max(hi,ends[i])
big=[g for g in segs if sum(1 for r in g if "k_dense_residual" in r["Kernel_Name"])>=2]
sel=big[-1] if big else max(segs,key=len)
t0=int(sel[0]["Start_Timestamp"])
# end of first step: the second k_dense_residual launch
cnt=0; tend=None
for r in sel:
    if "k_dense_residual" in r["Kernel_Name"]:
        cnt+=1
        if cnt==2: tend=int(r["Start_Timestamp"]); break
print("first step span us", (tend-t0)/1e3, "kernels", sum(1 for r in sel if int(r["Start_Timestamp"])<tend))
qs={}
prev_end={}
out=[]
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s>=tend: break
    q=r.get("Queue_Id","?")
    out.append(((s-t0)/1e3,(e-s)/1e3,q,r["Kernel_Name"].split("(")[0][:48]))
# summarize by 100-us windows which queue is busy, plus list kernels on the busiest queue
from collections import Counter
busy=Counter()
for t,d,q,nm in out: busy[q]+=d
print("busy us per queue", dict(busy))
mainq=max(busy,key=busy.get)
print("timeline (all queues):")
for t,d,q,nm in out:
    print(f"{t:9.1f} +{d:6.1f} q{q} {nm}")
PY
rm -rf gpurun_out/prof_t
