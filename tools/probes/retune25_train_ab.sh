# Race tile 25 against the train step's incumbents (eager engines autotune with GN_AUTOTUNE=1) and A/B the two tables on bench_train.py, alternating.
set -x
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
GN_AUTOTUNE=1 GN_RETUNE=25 python bench_train.py --steps 2 --warmup 1 > gpurun_out/r06_retune25_train_race.json 2> gpurun_out/r06_retune25_train_race.err
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json; cp /tmp/tune_new.json gpurun_out/gemm_tune_gfx950.json
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), 'loss', j.get('loss_last'), 'gnorm', j.get('grad_norm_last'))"; }
for i in 1 2; do
cp /tmp/tune_old.json genima_amd/gemm_tune_gfx950.json; python bench_train.py --steps 10 --warmup 3 2>/dev/null | p old
cp /tmp/tune_new.json genima_amd/gemm_tune_gfx950.json; python bench_train.py --steps 10 --warmup 3 2>/dev/null | p new
done
python - <<'P'
import json
old, new = json.load(open("/tmp/tune_old.json")), json.load(open("/tmp/tune_new.json"))
ch = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(len(ch), "table entries changed:")
for k, (a, b) in sorted(ch.items()): print("  ", k, a, "->", b)
P
