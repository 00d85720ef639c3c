set -x
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
python bench.py --steps 10 --warmup 3 > gpurun_out/ab_old1.json 2> gpurun_out/ab_old1.err
GN_RETUNE=15 python bench.py --steps 3 --warmup 1 > gpurun_out/retune.json 2> gpurun_out/retune.err
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json
python bench.py --steps 10 --warmup 3 > gpurun_out/ab_new1.json 2> gpurun_out/ab_new1.err
cp /tmp/tune_old.json genima_amd/gemm_tune_gfx950.json
python bench.py --steps 10 --warmup 3 > gpurun_out/ab_old2.json 2> gpurun_out/ab_old2.err
cp /tmp/tune_new.json genima_amd/gemm_tune_gfx950.json
python bench.py --steps 10 --warmup 3 > gpurun_out/ab_new2.json 2> gpurun_out/ab_new2.err
cp /tmp/tune_new.json gpurun_out/gemm_tune_new.json
python - <<'P'
import json
for n in ("old1","new1","old2","new2"):
    try:
        j=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j.get("train",{}).get("ms_per_step"), j.get("single_view_b1",{}).get("ms_per_call"))
    except Exception as e: print(n, "ERR", e)
P
