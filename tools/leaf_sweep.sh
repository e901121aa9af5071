#!/bin/bash
# usage (GPU box): bash tools/leaf_sweep.sh  -> gpurun_out/leaf_sweep.log   (DRE_LEAF_SIZE tuning knob of the nested dissection)
cd "$GRAFT_REPO_ROOT"
for leaf in 24 32 48 64 96 128; do
  for cfg in "5177 12" "20209 4"; do
    echo "leaf=$leaf cfg=$cfg"
    DRE_LEAF_SIZE=$leaf python tools/trace_one.py $cfg 2>&1 | tail -1
  done
done
