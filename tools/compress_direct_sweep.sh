set -e
for cfg in "512 4" "8192 4" "8192 2" "8192 8"; do
  set -- $cfg
  echo "== max_n=$1 ratio=$2"
  DRE_COMPRESS_DIRECT_MAX_N=$1 DRE_COMPRESS_DIRECT_RATIO=$2 timeout -k 10 200 python tools/profile_solve.py 1357 45 | grep -E "rep=|qr_panel|gemm_compress|gemm_band|total"
  DRE_COMPRESS_DIRECT_MAX_N=$1 DRE_COMPRESS_DIRECT_RATIO=$2 timeout -k 10 200 python tools/profile_solve.py 5177 10 | grep -E "rep=|qr_panel|gemm_compress|gemm_band|total"
done
