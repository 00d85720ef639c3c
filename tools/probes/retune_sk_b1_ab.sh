# deeper K splits (10 / 12 / 16 / 20 slices) raced against each table entry inside the B = 1 programs (GN_RETUNE_SK), then old table vs new table, alternating
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j.get('ms_per_call_median',0),3))"; }
F="--no-train --no-cpu-baseline --no-single-view --no-roofline"
T=genima_amd/gemm_tune_gfx950.json
cp $T /tmp/tune_old.json
GN_RETUNE_SK=10,12,16,20 python bench.py --workload single_b1 --steps 3 --warmup 1 $F > /dev/null 2>&1
GN_RETUNE_SK=10,12,16,20 python bench.py --workload tiled_b1 --steps 3 --warmup 1 $F > /dev/null 2>&1
cp $T /tmp/tune_new.json; cp $T gpurun_out/gemm_tune_resplit.json
python - <<'PY'
import json
a=json.load(open('/tmp/tune_old.json')); b=json.load(open('/tmp/tune_new.json'))
ch={k:(a[k],b[k]) for k in a if a[k]!=b.get(k)}
print(len(ch), "entries moved"); [print(k, v) for k, v in sorted(ch.items())]
PY
for i in 1 2; do for v in old new; do cp /tmp/tune_$v.json $T
python bench.py --workload tiled_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "tiled_b1 table=$v"
python bench.py --workload single_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "single_b1 table=$v"
python bench.py --steps 10 --warmup 3 $F 2>/dev/null | p "b8 table=$v"
done; done
cp /tmp/tune_old.json $T
