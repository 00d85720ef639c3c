# GN_ZERO_CONV_FUSED (ControlNet zero convs behind the join with the UNet skip as residual operand: no add launch) and GN_ZERO_CONV_SPLIT (dealt over
# both streams at small batch) A/B on the B = 8 call and the B = 1 tiled / single-view calls, alternating
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j.get('ms_per_call_median',0),3))"; }
F="--no-train --no-cpu-baseline --no-single-view --no-roofline"
for i in 1 2; do for v in "0 1" "1 0" "1 1"; do set -- $v
GN_ZERO_CONV_FUSED=$1 GN_ZERO_CONV_SPLIT=$2 python bench.py --steps 10 --warmup 3 $F 2>/dev/null | p "b8 fused=$1 split=$2"
GN_ZERO_CONV_FUSED=$1 GN_ZERO_CONV_SPLIT=$2 python bench.py --workload tiled_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "tiled_b1 fused=$1 split=$2"
GN_ZERO_CONV_FUSED=$1 GN_ZERO_CONV_SPLIT=$2 python bench.py --workload single_b1 --steps 20 --warmup 5 $F 2>/dev/null | p "single_b1 fused=$1 split=$2"
done; done
