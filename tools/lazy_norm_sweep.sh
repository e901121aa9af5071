for w in 0 1; do echo "== DRE_LAZY_NORM=$w"; DRE_LAZY_NORM=$w timeout -k 10 200 python tools/xevery_probe.py | cut -c1-330; done
