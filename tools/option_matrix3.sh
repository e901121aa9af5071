#!/bin/bash
# usage (GPU box): bash tools/option_matrix3.sh  -> gpurun_out/optmat3.log : the GDRE / GALE / boundary tests under the round-2b switches
cd "$GRAFT_REPO_ROOT"
for env in "DRE_SETUP_STREAMS=0" "DRE_MF_SUBTREE=1" "DRE_COMPRESS_SKETCH=0" "DRE_DENSE_X_MAX_N=0" "DRE_X_SIDE_STREAM=0" "DRE_FETCH_SPIN=0"; do
  echo "== $env"
  env $env timeout -k 10 600 python -m pytest tests/test_gpu_gdre.py tests/test_gpu_r02_configs.py tests/test_gpu_ldlt_gale.py tests/test_gpu_gare.py -x -q 2>&1 | tail -2
done
