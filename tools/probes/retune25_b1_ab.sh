# Race tile 25 inside the B = 1 calls (tiled observation, single view) and A/B the tables, alternating.
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j.get('ms_per_call_median',0),3))"; }
F="--no-train --no-cpu-baseline --no-single-view --no-roofline"
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
for w in tiled_b1 single_b1; do GN_RETUNE=25 python bench.py --workload $w --steps 2 --warmup 1 $F > /dev/null 2>&1; done
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json; cp /tmp/tune_new.json gpurun_out/gemm_tune_gfx950.json
for i in 1 2; do for t in old new; do
cp /tmp/tune_$t.json genima_amd/gemm_tune_gfx950.json
for w in tiled_b1 single_b1; do python bench.py --workload $w --steps 20 --warmup 5 $F 2>/dev/null | p "$w $t"; done
done; done
python - <<'P'
import json
old, new = json.load(open("/tmp/tune_old.json")), json.load(open("/tmp/tune_new.json"))
ch = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(len(ch), "table entries changed:")
for k, (a, b) in sorted(ch.items()): print("  ", k, a, "->", b)
P
