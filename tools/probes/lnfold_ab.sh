set -x
python -m pytest tests/test_gemm_tiles_gpu.py tests/test_models_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 > gpurun_out/lnf_tune.json 2> gpurun_out/lnf_tune.err
cp genima_amd/gemm_tune_gfx950.json gpurun_out/gemm_tune_lnf.json
for i in 1 2; do
GN_LN_FOLD=0 python bench.py --steps 10 --warmup 3 > gpurun_out/lnf_off$i.json 2> gpurun_out/lnf_off$i.err
python bench.py --steps 10 --warmup 3 > gpurun_out/lnf_on$i.json 2> gpurun_out/lnf_on$i.err
done
python - <<'P'
import json
for n in ("off1","on1","off2","on2"):
    try:
        j=json.loads(open(f"gpurun_out/lnf_{n}.json").read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j.get("train",{}).get("ms_per_step"), {k:v for k,v in j.get("single_view_b1",{}).items() if 'ms' in k})
    except Exception as e: print(n, "ERR", e)
P
