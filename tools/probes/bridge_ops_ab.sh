# per-(kernel, shape) tables of a workload with the GroupNorm bridge off / on (one box):  bash tools/probes/bridge_ops_ab.sh <tag> "<workloads>" "<extra env for on>"
TAG=${1:-r05}; WL=${2:-"tiled_b1 single_b1"}; EXTRA=${3:-}
for W in $WL; do
  for V in off on; do
    case $V in off) E="GN_BRIDGE=0";; on) E="GN_BRIDGE=1 $EXTRA";; esac
    env $E python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-single-view --no-act --dump-ops gpurun_out/${TAG}_ops_${W}_$V.csv 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $V', round(j['ms_per_step'],2))"
  done
done
