#!/bin/bash
# usage (GPU box): bash tools/subtree_level_sweep.sh  (root level of the subtree sweeps, sparse.hip SubPlan)
cd "$GRAFT_REPO_ROOT"
for lv in auto 5 6 7 8 9; do
  for cfg in "5177 12" "20209 4"; do
    if [ $lv = auto ]; then unset DRE_MF_SUBTREE_LEVEL; else export DRE_MF_SUBTREE_LEVEL=$lv; fi
    echo "level=$lv cfg=$cfg $(DRE_TRACE_SUBTREE=1 python tools/trace_one.py $cfg 2>&1 | grep -E 'subtree sweeps|rep=2' | tr '\n' ' ')"
  done
done
