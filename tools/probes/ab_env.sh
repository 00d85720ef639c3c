# same-box alternating A/B of one environment switch over the driver-like bench:  VAR=GN_TAP_MINOR bash tools/probes/ab_env.sh [reps]
# prints ms per B=8 call, single-view B=1 and tiled B=1 medians for VAR=1 / VAR=0
VAR=${VAR:-GN_TAP_MINOR}; REPS=${1:-2}
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), j.get('single_view_b1',{}).get('ms_per_call_median'), j.get('tiled_b1',{}).get('ms_per_call_median'))"; }
for i in $(seq $REPS); do
  env $VAR=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline ${BENCH_EXTRA} 2>/dev/null | p "$VAR=1"
  env $VAR=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-roofline ${BENCH_EXTRA} 2>/dev/null | p "$VAR=0"
done
