# per-(kernel, shape) tables of workloads with one environment switch on / off (one box):  VAR=GN_REDUCE_FUSE bash tools/probes/env_ops_ab.sh <tag> "<workloads>"
TAG=${1:-r05}; WL=${2:-"tiled_b8 tiled_b1"}; VAR=${VAR:-GN_REDUCE_FUSE}
for W in $WL; do
  for V in 0 1; do
    env $VAR=$V python bench.py --workload $W --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-single-view --no-act --dump-ops gpurun_out/${TAG}_ops_${W}_${VAR}$V.csv 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $VAR=$V', round(j['ms_per_step'],2))"
  done
done
