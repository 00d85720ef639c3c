# what, of the things bench.py's extras leave behind in the process, slows the in-process train extra when it runs LAST (66 vs 63 ms)?
# one box; GN_BENCH_TRAIN=last; each line = one bench.py run with some extras switched off
run() { env GN_BENCH_TRAIN=last "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['train']['ms_per_step'],2))"; }
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
echo "all extras:                         $(run $B)"
echo "no roofline replay:                 $(run $B --no-roofline)"
echo "no b1 / two-calls extras:           $(run $B --no-single-view)"
echo "neither:                            $(run $B --no-roofline --no-single-view)"
echo "no hipGraph captures (b1 extras):   $(GN_BENCH_SKIP=graph run $B)"
echo "no two-calls-in-flight:             $(GN_BENCH_SKIP=two_calls run $B)"
echo "no graphs, no two-calls:            $(GN_BENCH_SKIP=graph,two_calls run $B)"
echo "no graphs, no two-calls, no replay: $(GN_BENCH_SKIP=graph,two_calls run $B --no-roofline)"
