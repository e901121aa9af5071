#!/bin/bash
# usage (GPU box): bash tools/mf_latency_sweep.sh   -> mf_solve_real time per configuration of the latency/throughput kernel split
cd "$GRAFT_REPO_ROOT"
for cfg in "20209 4" "5177 12"; do
  for fw in 0 256 700 1500; do
    for bw in 0 700 1500 3000; do
      r=$(DRE_MF_LAT_FWD=$fw DRE_MF_LAT_BWD=$bw python tools/profile_solve.py $cfg 2>&1 | grep -E "rep=1|mf_solve_real" | awk '{printf "%s ", $0}')
      echo "fwd<=$fw bwd<=$bw :: $r"
    done
  done
done
