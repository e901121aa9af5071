#!/bin/bash
# Kernel table of any python tool under rocprofv3 (GPU box): bash tools/prof_any.sh <rows> <script.py> [args...]
rows=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_A
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_A -- python "$@" > gpurun_out/prof_any_run.log 2>&1
f=$(find gpurun_out/prof_A -name "*kernel_stats.csv" | head -1)
python - "$f" "$rows" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:int(sys.argv[2])]:
    print("%-72s %6s %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf gpurun_out/prof_A
