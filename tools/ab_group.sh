#!/bin/bash
# A/B on the metric's configuration (run on the GPU box): usage tools/ab_group.sh "ENV=.. ENV=.." ...   (default: group chain off / auto)
set -e
cd "$GRAFT_REPO_ROOT"
[ $# -eq 0 ] && set -- "DRE_OPTIONS=adi_group=0" "DRE_OPTIONS=adi_group=1"
python -m pytest tests/test_gpu_r03_full_length.py -x -q -m gpu -k "metric or falls_back or group" > gpurun_out/ab_t.log 2>&1 || { tail -30 gpurun_out/ab_t.log; exit 1; }
tail -1 gpurun_out/ab_t.log
for cfg in "$@"; do
  for rep in 1 2; do
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-general-path > gpurun_out/ab_g.json 2> gpurun_out/ab_g.err || { tail -5 gpurun_out/ab_g.err; exit 1; }
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_g.json"))
print(sys.argv[1], round(d["value"]), "it/s", round(d["ms_per_step"],2), "ms", d["config"]["adi_iterations_per_solve"], d["config"]["parity"]["delta_K_worst"], {k:v for k,v in list(d["roofline"]["by_kernel_ms"].items())[:7]})
PY
  done
done
env DRE_TRACE=phase python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-general-path 2>&1 >/dev/null | grep "wall, ms" | tail -2
