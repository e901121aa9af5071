for rep in 1 2; do for w in 513 64; do echo "== lead rotation min n $w"
DRE_LEAD_ROTATION_MIN_N=$w timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
