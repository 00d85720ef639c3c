# same-box A/B of the train step under environment switches:  bash tools/probes/train_ab.sh "GN_DEFER_RELEASE=0" "GN_WGRAD_BLOCKS=512" ...
# (every argument is one variant's environment, "" = defaults; two alternating rounds)
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), 'loss', j['loss_last'], 'gnorm', j['grad_norm_last'])"; }
for r in 1 2; do
  for v in "$@"; do
    env $v python bench_train.py --steps 8 --warmup 3 2>/dev/null | p "[$v]"
  done
done
