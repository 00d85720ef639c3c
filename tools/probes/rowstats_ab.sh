p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('train',{}).get('ms_per_step'), j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
python -m pytest tests/test_gemm_tiles_gpu.py tests/test_models_gpu.py -x -q 2>&1 | grep "passed\|failed"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp genima_amd/gemm_tune_gfx950.json gpurun_out/gemm_tune_rs.json
for i in 1 2; do
GN_ROW_STATS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "rs0"
GN_ROW_STATS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | p "rs1"
done
