#!/bin/bash
# The GPU suite under the non-default code paths (run on the GPU box): every configuration must stay green.
cd "$GRAFT_REPO_ROOT"
rc=0
for cfg in "DRE_ADI_GROUP=0" "DRE_ADI_GROUP=2" "DRE_X_SIDE_STREAM=0" "DRE_DENSE_X_MAX_N=0" "DRE_SETUP_STREAMS=0 DRE_SIDE_EARLY=1" "DRE_TSMM=1 DRE_TOP_FUSED=8" \
           "DRE_ADI_FAN=0 DRE_PREFETCH_FACTORS=0" "DRE_ADI_FAN=4 DRE_ADI_FAN_MAX_COEF=8 DRE_PREFETCH_FACTORS=3 DRE_STREAM_PRIORITIES=1" "DRE_DENSE_INV_MAX_N=0 DRE_DENSE_X_MAX_N=0 DRE_ADI_FAN=4"; do
  env $cfg python -m pytest tests -x -q -m gpu > gpurun_out/om.log 2>&1
  r=$?
  echo "$cfg -> rc=$r  $(tail -1 gpurun_out/om.log)"
  [ $r -ne 0 ] && { rc=1; tail -25 gpurun_out/om.log; }
done
exit $rc
