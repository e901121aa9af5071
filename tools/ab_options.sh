#!/bin/bash
# A/B of option sets on ONE box, interleaved: tools/ab_options.sh <rounds> <n> <steps> <reps> "<opts A>" "<opts B>" ...   (run on the GPU box)
rounds=$1; n=$2; steps=$3; reps=$4; shift 4
for r in $(seq 1 $rounds); do
  for o in "$@"; do
    DRE_OPTIONS="$o" timeout -k 10 200 python tools/time_headline.py $n $steps $reps 2>&1 | grep median
  done
done
