#!/bin/bash
# A/B of options on one configuration (GPU box): usage tools/ab_env.sh "<bench args>" "DRE_OPTIONS=name=value,.." ...   (interleaved, 2 rounds)
cd "$GRAFT_REPO_ROOT"
bargs=$1; shift
for round in $(seq ${AB_ROUNDS:-2}); do
for cfg in "$@"; do
    env $cfg python bench.py $bargs --no-cpu-baseline --no-general-path > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || { tail -5 gpurun_out/ab_env.err; continue; }
    python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_env.json"))
print(sys.argv[1], round(d["value"]), "it/s", round(d["ms_per_step"],2), "ms", d["config"]["adi_iterations_per_solve"], d["config"].get("parity") and d["config"]["parity"]["delta_K_worst"], {k:v for k,v in list(d["roofline"]["by_kernel_ms"].items())[:6]})
PY
done; done
