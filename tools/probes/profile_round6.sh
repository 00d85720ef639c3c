# Round-6 evidence run: as profile_round5b.sh (bench line, rocprofv3 kernel stats + exact-N-call window, per-shape op tables,
# train step, attention micro-benchmark, PMC traffic stamped with GIT_COMMIT); summaries are copied to profiles/r06_v<N>_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o b8 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-single-view > $O/inf.log 2>&1
python tools/probes/trace_window.py $O/inf 300 3 image_f16_to_u8_kernel > $O/inf_window.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t8 -- python bench_train.py --steps 5 --warmup 2 > $O/train.log 2>&1
python tools/probes/trace_window.py $O/train 360 4 adamw_kernel > $O/train_window.txt
python bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/train_bench.json
python bench.py --dump-ops $O/ops_b8.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python bench.py --workload tiled_b1 --dump-ops $O/ops_tiled_b1.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python bench.py --workload single_b1 --dump-ops $O/ops_b1.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python tools/bench_attn.py 2>/dev/null > $O/attn.txt
(python tools/probes/ppp_ksweep.py 2>&1 | grep -v amdgpu.ids) > $O/ppp_ksweep.txt
rm -f $O/inf/*kernel_trace.csv $O/train/*kernel_trace.csv $O/inf/*/*kernel_trace.csv $O/train/*/*kernel_trace.csv
bash tools/probes/pmc_traffic.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_traffic/traffic.json $O/pmc_traffic.json
find $O -name "*stats.csv" | head; tail -c 300 $O/bench_full.json | head -c 250; echo; cut -c1-160 $O/train_bench.json; tail -3 $O/pmc.log
