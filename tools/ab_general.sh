#!/bin/bash
# A/B of general-path switches at n = 5177 (12 steps) and n = 20209 (12 steps); run on the GPU box.  usage: tools/ab_general.sh "DRE_OPTIONS=name=value,.." ...
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  for nn in "5177 3" "20209 2"; do
    set -- $nn
    env $cfg python bench.py --n $1 --nsteps 12 --steps $2 --warmup 1 --no-cpu-baseline --no-general-path > gpurun_out/ab_gen.json 2> gpurun_out/ab_gen.err || { tail -5 gpurun_out/ab_gen.err; continue; }
    python - "$cfg" $1 <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_gen.json"))
print(sys.argv[1], "n", sys.argv[2], round(d["value"]), "it/s", round(d["ms_per_step"],1), "ms", d["config"]["adi_iterations_per_solve"], d["config"].get("parity") and d["config"]["parity"]["delta_K_worst"], {k:v for k,v in list(d["roofline"]["by_kernel_ms"].items())[:8]})
PY
  done
done
