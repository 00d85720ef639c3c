# same-box alternating A/B of two tune tables over the driver-like bench: TABLE_B=<json> bash tools/probes/ab_tune_table.sh [reps]   (A = the shipped table)
REPS=${1:-2}; A=genima_amd/gemm_tune_gfx950.json; cp $A /tmp/tune_A.json
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), j.get('single_view_b1',{}).get('ms_per_call_median'), j.get('tiled_b1',{}).get('ms_per_call_median'))"; }
for i in $(seq $REPS); do
  cp /tmp/tune_A.json $A; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>/dev/null | p "table A (shipped)"
  cp $TABLE_B $A;         python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>/dev/null | p "table B"
done
cp /tmp/tune_A.json $A
