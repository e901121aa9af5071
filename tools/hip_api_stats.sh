#!/bin/bash
# host-side HIP API statistics of one solve loop (tools/trace_one.py n steps): how much of the wall clock is launch / sync / event calls.   (run on the GPU box)
n=$1; steps=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_h
timeout -k 10 400 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/prof_h -- python tools/trace_one.py $n $steps > gpurun_out/hip_api_$n.log 2> gpurun_out/prof_h.err
f=$(find gpurun_out/prof_h -name "*hip_api_stats.csv" | head -1)
echo "stats file: $f"
head -25 "$f"
rm -rf gpurun_out/prof_h
