#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
for cfg in "1 768" "1 1536"; do
  set -- $cfg
  DRE_ADI_GROUP=$1 DRE_ADI_GROUP_MAX_N=$2 python bench.py --n 1357 --steps 3 --warmup 1 --no-cpu-baseline --no-general-path > gpurun_out/ab_g.json 2> gpurun_out/ab_g.err || { tail -5 gpurun_out/ab_g.err; exit 1; }
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_g.json"))
print("group/maxn", sys.argv[1], round(d["value"]), "it/s", round(d["ms_per_step"],2), "ms", d["config"]["adi_iterations_per_solve"], d["config"]["parity"], {k:v for k,v in list(d["roofline"]["by_kernel_ms"].items())[:7]}, d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3))
PY
done
