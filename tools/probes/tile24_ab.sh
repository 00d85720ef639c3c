set -x
mkdir -p gpurun_out/t24
timeout 900 python -m pytest tests/test_gemm_tiles_gpu.py tests/test_k_append_gpu.py tests/test_upsample_phases_gpu.py tests/test_program_plan_gpu.py -x -q -m gpu 2>&1 | tail -5
CFGS=12,22,23,19,6 B=8 timeout 600 python tools/bench_gemm.py > gpurun_out/t24/gemm_tiles_b8.txt 2>&1
cat gpurun_out/t24/gemm_tiles_b8.txt | cut -c1-330
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), j.get('single_view_b1',{}).get('ms_per_call_median'), j.get('tiled_b1',{}).get('ms_per_call_median'))"; }
for i in 1 2; do
  GN_RETUNE=24 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>gpurun_out/t24/retune_$i.err | p "retune24"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>/dev/null | p "base"
done
grep -i "retune\|-> 24\|tile 24" gpurun_out/t24/retune_1.err | head -40
