p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j.get('ms_per_call_median',0),3))"; }
F="--no-train --no-cpu-baseline --no-single-view --no-roofline"
for i in 1 2; do for v in 0 1 2 4; do
GN_PROBE_SPLITK_CAP=$v python bench.py --workload tiled_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "tiled_b1 cap=$v"
GN_PROBE_SPLITK_CAP=$v python bench.py --workload single_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "single_b1 cap=$v"
done; done
