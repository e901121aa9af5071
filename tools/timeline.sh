#!/bin/bash
# usage (GPU box): bash tools/timeline.sh <n> <nsteps> <tag>   -> gpurun_out/timeline_<tag>.txt
n=${1:-371}; steps=${2:-45}; tag=${3:-t}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python tools/trace_one.py $n $steps > gpurun_out/timeline_$tag.log 2>&1
f=$(ls gpurun_out/prof_tl/*/*kernel_trace.csv | head -1)
TIMELINE_HEAD=${TIMELINE_HEAD:-0} python tools/timeline.py "$f" $steps > gpurun_out/timeline_$tag.txt 2>&1
rm -rf gpurun_out/prof_tl
cat gpurun_out/timeline_$tag.log | tail -4
cat gpurun_out/timeline_$tag.txt
