set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; }
run DRE_X_SIDE_STREAM=0
run DRE_X_SIDE_STREAM=0 DRE_X_COMPRESS_EVERY=3
run DRE_LAZY_NORM=0
run DRE_FOLD_E=0
run DRE_XBLOCKS_MAX_N=512
