p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j.get('single_view_b1',{}).get('ms_per_call_median'))"; }
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
python - <<'P'
import json
p='genima_amd/gemm_tune_gfx950.json'
t=json.load(open(p)); n=len(t)
t={k:v for k,v in t.items() if not k.endswith('|ln')}
json.dump(dict(sorted(t.items())), open(p,'w'), indent=0); print('stripped', n-len(t))
P
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2>&1
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json; cp /tmp/tune_new.json gpurun_out/gemm_tune_ln2.json
for i in 1 2; do
cp /tmp/tune_old.json genima_amd/gemm_tune_gfx950.json; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | p old
cp /tmp/tune_new.json genima_amd/gemm_tune_gfx950.json; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | p new
done
