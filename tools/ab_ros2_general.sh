#!/bin/bash
# A/B of option sets on the Ros2 general-path run, interleaved on ONE box: tools/ab_ros2_general.sh <rounds> "<opts A>" "<opts B>" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for o in "$@"; do
    echo -n "options='$o' "; DRE_OPTIONS="$o" timeout -k 10 200 python tools/time_ros2_general.py 5177 12 3 conv 2>&1 | head -1 | sed 's/.*converged/converged/'
  done
done
