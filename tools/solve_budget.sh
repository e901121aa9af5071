#!/bin/bash
# rocprofv3 kernel trace of bench.py (n = 371): per-kernel totals over the LAST solve, split by queue, + per-step kernel counts (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> gpurun_out/prof_t.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_dense_residual" in r["Kernel_Name"]][-44:]
t0=int(rows[idx[0]]["Start_Timestamp"])-30000
sel=[r for r in rows if int(r["Start_Timestamp"])>=t0]
t1=max(int(r["End_Timestamp"]) for r in sel)
print("span of steps 2..45 (us):", (t1-t0)/1e3)
tot=defaultdict(lambda:[0,0.0]); 
qb=defaultdict(float)
for r in sel:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    k=(r["Queue_Id"], r["Kernel_Name"].split('(')[0].replace("void ","")[:40])
    tot[k][0]+=1; tot[k][1]+=d; qb[r["Queue_Id"]]+=d
print("busy per queue:", dict(qb))
for k,v in sorted(tot.items(), key=lambda kv:-kv[1][1]):
    print(f"q{k[0]} {k[1]:42s} n={v[0]:5d} total={v[1]:9.1f} us  avg={v[1]/v[0]:6.2f}")
PY
rm -rf gpurun_out/prof_t
