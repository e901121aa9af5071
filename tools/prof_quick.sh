#!/bin/bash
# quick rocprofv3 kernel stats of bench.py (GPU box): usage tools/prof_quick.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_q
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-path "$@" > gpurun_out/prof_q_$tag.json 2> gpurun_out/prof_q_$tag.err
f=$(find gpurun_out/prof_q -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/prof_q_$tag.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:45]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} tot_ms {float(r["TotalDurationNs"])/1e6:8.3f} avg_us {float(r["AverageNs"])/1e3:8.2f} {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
rm -rf gpurun_out/prof_q
cat gpurun_out/prof_q_$tag.txt
