# same-box alternating A/B of attention_pwg.hip's routing over the driver-like bench: GN_ATTN_PWG_MIN_KEYS=2048 (default: on) vs 0 (off)
REPS=${1:-2}
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), j.get('single_view_b1',{}).get('ms_per_call_median'), j.get('tiled_b1',{}).get('ms_per_call_median'))"; }
for i in $(seq $REPS); do
  env GN_ATTN_PWG_MIN_KEYS=2048 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>/dev/null | p "pwg on "
  env GN_ATTN_PWG_MIN_KEYS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline 2>/dev/null | p "pwg off"
done
