p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('train',{}).get('ms_per_step'), j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
true
true
cp genima_amd/gemm_tune_gfx950.json gpurun_out/gemm_tune_vrow.json
for i in 1 2; do
GN_ROWMAJOR_V=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "vrow0"
GN_ROWMAJOR_V=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "vrow1"
done
