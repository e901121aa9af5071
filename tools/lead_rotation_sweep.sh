for w in 0 1; do echo "== lead rotation $w"
DRE_LEAD_ROTATION=$w timeout -k 10 200 python tools/profile_solve.py 371 45 | grep -E "rep=|total"
DRE_LEAD_ROTATION=$w timeout -k 10 200 python tools/profile_solve.py 1357 45 | grep -E "rep=|total"
DRE_LEAD_ROTATION=$w timeout -k 10 200 python tools/factor_vs_qr.py 1357 | grep -E "new|qr|diff"
done
