# train step: eager tape against the captured hipGraph of the forward + backward walk (GN_TRAIN_GRAPH), alternating
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3))"; }
for i in 1 2; do
python bench_train.py --steps 10 --warmup 4 2>/dev/null | p "train eager"
python bench_train.py --steps 10 --warmup 4 --graph 2>/dev/null | p "train graph"
done
