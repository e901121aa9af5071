#!/bin/bash
# Kernel table of the Ros2 general-path run (GPU box): rocprofv3 --kernel-trace --stats around tools/time_ros2_general.py.   usage: bash tools/prof_ros2_general.sh [steps] [conv|s12]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_R -- python tools/time_ros2_general.py 5177 ${1:-12} 1 ${2:-conv} > gpurun_out/prof_ros2_general_run.log 2>&1
f=$(find gpurun_out/prof_R -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms (two solves: warm-up + timed)" % (tot / 1e6))
for r in rows[:36]:
    print("%-72s %6s %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -1 gpurun_out/prof_ros2_general_run.log | cut -c1-300
rm -rf gpurun_out/prof_R
