#!/bin/bash
# usage (GPU box): bash tools/mf_levels.sh <n> <nsteps> <tag> [env assignments...]
n=$1; steps=$2; tag=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/prof_ml
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ml -- python tools/trace_one.py $n $steps > gpurun_out/mf_levels_$tag.log 2>&1
f=$(ls gpurun_out/prof_ml/*/*kernel_trace.csv | head -1)
python tools/mf_levels.py "$f" > gpurun_out/mf_levels_$tag.txt 2>&1
rm -rf gpurun_out/prof_ml
