p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('train',{}).get('ms_per_step'), j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
for i in 1 2; do
GN_LN_FOLD_GEGLU=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | p "geglu-fold on "
GN_LN_FOLD_GEGLU=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | p "geglu-fold off"
done
