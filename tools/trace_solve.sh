#!/bin/bash
# rocprofv3 kernel trace of tools/profile_solve.py; prints the per-kernel table and busy/span of the last solve.
# usage (on the GPU box): bash tools/trace_solve.sh <n> <nsteps> <tag>
n=${1:-371}; steps=${2:-45}; tag=${3:-trace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_k
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python tools/profile_solve.py $n $steps > gpurun_out/prof_k_$tag.log 2>&1
python tools/profile_summarize.py $tag gpurun_out/prof_k gpurun_out/sum_$tag > /dev/null 2>&1
f=$(ls gpurun_out/prof_k/*/*kernel_trace.csv | head -1)
python - "$f" <<"PY"
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); seg = rows[2 * n // 3:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
print("last third of the trace: kernels", len(seg), "busy ms %.1f" % (busy / 1e6), "span ms %.1f" % (span / 1e6), "median gap us %.2f" % (statistics.median(gaps) / 1e3))
PY
rm -rf gpurun_out/prof_k
head -34 gpurun_out/sum_$tag/rocprof_${tag}_kernel_stats.md | cut -c1-150
