for w in 0 1; do echo "== defer $w"
DRE_DEFER_COMPRESS=$w timeout -k 10 200 python tools/profile_solve.py 1357 45 | grep -E "rep=|total"
DRE_DEFER_COMPRESS=$w timeout -k 10 200 python tools/profile_solve.py 5177 45 | grep -E "rep=|total"
DRE_DEFER_COMPRESS=$w timeout -k 10 300 python tools/profile_solve.py 20209 4 | grep -E "rep=|total"
done
