for w in 512 1536; do echo "== DRE_XBLOCKS_MAX_N=$w"; DRE_XBLOCKS_MAX_N=$w timeout -k 10 200 python tools/profile_solve.py 1357 45 | grep -E "rep=|total"; done
