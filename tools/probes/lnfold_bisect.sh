p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('train',{}).get('ms_per_step'), j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
GN_LN_FOLD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fold1 noCPU"
GN_LN_FOLD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fold0 noCPU"
GN_LN_FOLD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fold1 noCPU"
GN_LN_FOLD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fold0 noCPU"
