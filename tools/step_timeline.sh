#!/bin/bash
# rocprofv3 kernel trace of bench.py (n = 371): one steady-state time step (the 10th of the last solve) — kernels per queue with gaps (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-path > /dev/null 2> gpurun_out/prof_t.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" ${1:-10} <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); which=int(sys.argv[2])
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_dense_residual" in r["Kernel_Name"]]
# last solve = last 45 residual launches
idx=idx[-45:]
a=idx[which-1]; b=idx[which]
# step boundaries: from the transpose before residual... take [start of k_dense_residual(which) - 40us, next]
t0=int(rows[a]["Start_Timestamp"])-30000; t1=int(rows[b]["Start_Timestamp"])-30000
sel=[r for r in rows if t0<=int(r["Start_Timestamp"])<t1]
print("step span us", (t1-t0)/1e3, "kernels", len(sel))
from collections import defaultdict
busy=defaultdict(float)
for r in sel: busy[r["Queue_Id"]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print("busy per queue", dict(busy))
mainq=max(busy,key=busy.get)
prev_end=None
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"]); q=r["Queue_Id"]
    gap=""
    if q==mainq:
        if prev_end is not None and s-prev_end>8000: gap=f"   <== main idle {(s-prev_end)/1e3:.1f} us"
        prev_end=e
    print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:6.1f} q{q} {r['Kernel_Name'].split('(')[0][:44]}{gap}")
PY
rm -rf gpurun_out/prof_t
