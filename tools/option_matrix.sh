#!/bin/bash
# The GPU suite under the non-default code paths (run on the GPU box): every configuration must stay green.
# Options travel as DRE_OPTIONS="name=value,..." (dre_ctx_set_option for every context of the process); there are no kernel-selecting
# environment variables any more (round 4).
cd "$GRAFT_REPO_ROOT"
rc=0
for cfg in "adi_group=0" "adi_group=2" "x_side_stream=0" "dense_x_max_n=0" "setup_streams=0" "adi_fan=0" "adi_fan=8,adi_fan_max_coef=8" \
           "ros1_recurrence=0" "dense_inverse_max_n=0,dense_x_max_n=0,adi_fan=4" "compress_sketch=0" "top_inverse_max_rows=0" \
           "setup_batched=0,side_after_panels=-1" "prefetch_batch=0" "dense_warm=0" "dense_warm=2" "recurrence_wide=0" "comm_host_async=1,side_gate=1" \
           "side_prefetch=1" "gemm_swizzle=0,mf_swizzle=1" "ros2_tight=0" "ros2_tight=2"; do
  DRE_OPTIONS="$cfg" python -m pytest tests -x -q -m gpu > gpurun_out/om.log 2>&1
  r=$?
  echo "$cfg -> rc=$r  $(tail -1 gpurun_out/om.log)"
  [ $r -ne 0 ] && { rc=1; tail -25 gpurun_out/om.log; }
done
exit $rc
