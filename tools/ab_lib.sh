#!/bin/bash
# A/B of two builds of libdre_hip.so on the metric configuration (GPU box): interleaved bench.py runs, minimum and median ms per solve.
# usage: tools/ab_lib.sh <libA> <libB> [rounds] [bench args]      (libA / libB: paths, "cur" = the in-tree build)
A=$1; B=$2; R=${3:-5}; shift 3
run() { lib=$1; shift; if [ "$lib" = cur ]; then unset DRE_HIP_LIB; else export DRE_HIP_LIB=$lib; fi
  python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-general-path "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in $(seq $R); do a="$a $(run $A "$@")"; b="$b $(run $B "$@")"; done
python - "$a" "$b" <<'PY'
import sys,statistics as st
for name,v in zip("AB",sys.argv[1:3]):
    x=sorted(float(t) for t in v.split()); print(name,"min %.3f median %.3f max %.3f ms per solve"%(x[0],st.median(x),x[-1]))
PY
