for w in 0 1536 2560; do echo "== top_inverse_max_rows $w"
DRE_TOP_INVERSE_MAX_ROWS=$w timeout -k 10 200 python tools/profile_solve.py 5177 10 | grep -E "rep=|mf_|total"
DRE_TOP_INVERSE_MAX_ROWS=$w timeout -k 10 300 python tools/profile_solve.py 20209 4 | grep -E "rep=|mf_|total"
done
