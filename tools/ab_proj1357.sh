#!/bin/bash
# A/B of option sets on BASELINE configs[2] (n = 1357 Ros2, Projection(2)), interleaved on ONE box: tools/ab_proj1357.sh <rounds> "<opts A>" "<opts B>" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for o in "$@"; do
    DRE_OPTIONS="$o" timeout -k 10 200 python tools/time_proj1357.py 2 2>&1 | head -3 | cut -c1-400
  done
done
