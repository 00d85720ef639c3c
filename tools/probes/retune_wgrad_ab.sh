# the eight-wave weight-gradient tiles (3 = 128 x 256, 4 = 256 x 128) raced into the train step (GN_RETUNE_WGRAD), then old table vs new table, alternating
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3))"; }
T=genima_amd/gemm_tune_gfx950.json
python -m pytest tests/test_backward_gpu.py -q -m gpu -k wgrad 2>&1 | tail -3
python tools/probes/wgrad_bench.py 2>/dev/null | grep -v amdgpu
cp $T /tmp/tune_old.json
GN_RETUNE_WGRAD=3,4 python bench_train.py --steps 2 --warmup 1 > /dev/null 2>&1
cp $T /tmp/tune_new.json; cp $T gpurun_out/gemm_tune_wgrad.json
python - <<'PY'
import json
a=json.load(open('/tmp/tune_old.json')); b=json.load(open('/tmp/tune_new.json'))
ch={k:(a[k],b[k]) for k in a if a[k]!=b.get(k)}
print(len(ch), "entries moved"); [print(k, v) for k, v in sorted(ch.items())]
PY
for i in 1 2; do for v in old new; do cp /tmp/tune_$v.json $T
python bench_train.py --steps 10 --warmup 3 2>/dev/null | p "train table=$v"
done; done
cp /tmp/tune_old.json $T
