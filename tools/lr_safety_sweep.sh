for w in 0 1; do echo "== warm $w"; DRE_LR_WARM=$w timeout -k 10 200 python tools/factor_vs_qr.py 5177 | grep -E "new|qr|diff"; done
