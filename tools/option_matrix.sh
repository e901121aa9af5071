# the whole GPU suite under the non-default code paths (each line: environment overrides)
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; }
run DRE_LEAD_ROTATION=0
run DRE_TOP_INVERSE_MAX_ROWS=0
run DRE_DEFER_COMPRESS=0
run DRE_COMPRESS_FACTOR_MIN_N=1073741824 DRE_COMPRESS_DIRECT_MAX_N=512
run DRE_DENSE_INV_MAX_N=0
run DRE_MF_SCALAR=1
