# Round-2 evidence run, final pass (LayerNorm fold, ping-pong K splits, attention backward occupancy): the default bench line, rocprofv3
# kernel stats of the same commands, per-shape op table, train-step window, PMC traffic.  Summaries are copied to profiles/r02_v6_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02c; rm -rf $O; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o b8 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-single-view > $O/inf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t8 -- python bench_train.py --steps 5 --warmup 2 > $O/train.log 2>&1
python tools/probes/trace_window.py $O/train 360 5 > $O/train_window.txt
python tools/probes/trace_window.py $O/inf 300 3 > $O/inf_window.txt
python bench_train.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/train_bench.json
python bench_train.py --family sdxl-turbo 2>/dev/null | tail -1 > $O/train_sdxl.json
python bench_train.py --family sdxl-turbo --fp8 2>/dev/null | tail -1 > $O/train_sdxl_fp8.json
python bench.py --dump-ops $O/ops_b8.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python bench.py --workload single_b1 --dump-ops $O/ops_b1.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python tools/bench_attn.py 2>/dev/null > $O/attn.txt
python tools/probes/gemm_ksweep.py 2>/dev/null | grep "^M=" > $O/gemm_ksweep.txt
python tools/probes/gn_bench.py 2>/dev/null | grep groupnorm > $O/gn_bench.txt
python tools/probes/wgrad_bench.py 2>/dev/null | grep "^linear\|^conv" > $O/wgrad_bench.txt
find $O -name "*stats.csv" | head; rm -f $O/inf/*kernel_trace.csv $O/train/*kernel_trace.csv $O/inf/*/*kernel_trace.csv $O/train/*/*kernel_trace.csv
bash tools/probes/pmc_traffic.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $O/pmc_traffic.json
tail -c 400 $O/bench_full.json | head -c 300; echo; cut -c1-160 $O/train_bench.json; cut -c1-160 $O/train_sdxl.json; cut -c1-160 $O/train_sdxl_fp8.json; tail -3 $O/pmc.log
