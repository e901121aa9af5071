#!/bin/bash
# A/B of the group chain (DRE_ADI_GROUP=0: one launch per ADI iteration) on the metric's configuration; run on the GPU box
set -e
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_r03_full_length.py -x -q -m gpu -k "metric or falls_back" > gpurun_out/ab_t.log 2>&1 || { tail -30 gpurun_out/ab_t.log; exit 1; }
tail -2 gpurun_out/ab_t.log
for cfg in "0 -1" "1 -1" "1 0" "2 -1"; do
  set -- $cfg
  for rep in 1 2; do
  DRE_ADI_GROUP=$1 DRE_SIDE_EARLY=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-general-path > gpurun_out/ab_g.json 2> gpurun_out/ab_g.err || { tail -5 gpurun_out/ab_g.err; exit 1; }
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_g.json"))
print("group/early", sys.argv[1], round(d["value"]), "it/s", round(d["ms_per_step"],2), "ms", d["config"]["adi_iterations_per_solve"], d["config"]["parity"]["delta_K_worst"], {k:v for k,v in list(d["roofline"]["by_kernel_ms"].items())[:7]})
PY
  done
done
