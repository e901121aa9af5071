#!/bin/bash
# usage (GPU box): bash tools/sketch_probe.sh -> gpurun_out/sketch_probe.log  (randomized wide-factor compression on / off at n = 5177, 20209)
cd "$GRAFT_REPO_ROOT"
for sk in 1 0; do
  for cfg in "5177 12" "20209 4"; do
    echo "sketch=$sk cfg=$cfg"
    DRE_COMPRESS_SKETCH=$sk DRE_TRACE_COMPRESS=1 python tools/trace_one.py $cfg 2>&1 | grep -E "sketch|rep=" | tail -12
  done
done
