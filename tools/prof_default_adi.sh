#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -- python tools/time_default_adi.py ${1:-371} ${2:-10} ${3:-1} > gpurun_out/prof_d.json 2> gpurun_out/prof_d.err
f=$(find gpurun_out/prof_d -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:45]:
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):6d} tot_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.2f} {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
cat gpurun_out/prof_d.json
rm -rf gpurun_out/prof_d
