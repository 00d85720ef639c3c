p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('train',{}).get('ms_per_step'), j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
GN_SPLITK_FIXUP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fix0 oldtable"
GN_SPLITK_FIXUP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fix1 oldtable"
GN_SPLITK_FIXUP=1 GN_RETUNE=10,11,17,18 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json; cp /tmp/tune_new.json gpurun_out/gemm_tune_skfix.json
GN_SPLITK_FIXUP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fix1 newtable"
cp /tmp/tune_old.json genima_amd/gemm_tune_gfx950.json
GN_SPLITK_FIXUP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fix0 oldtable"
cp /tmp/tune_new.json genima_amd/gemm_tune_gfx950.json
GN_SPLITK_FIXUP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "fix1 newtable"
