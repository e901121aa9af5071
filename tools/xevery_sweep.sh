for cfg in "DRE_X_COMPRESS_EVERY=1" "DRE_X_COMPRESS_EVERY=3" "DRE_X_SIDE_STREAM=1"; do echo "== $cfg"; env $cfg timeout -k 10 200 python tools/xevery_probe.py; done
