cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_v7; mkdir -p $O
cd $R
python bench.py > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o b8 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/inf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t8 -- python bench_train.py --steps 5 --warmup 2 > $O/train.log 2>&1
python bench_train.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/train_bench.json
python bench_train.py --family sdxl-turbo 2>/dev/null | tail -1 > $O/train_sdxl.json
python bench_train.py --family sdxl-turbo --fp8 2>/dev/null | tail -1 > $O/train_sdxl_fp8.json
python bench.py --workload single_b1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b1.json || true
python tools/bench_attn.py 2>/dev/null > $O/attn.txt
python tools/bench_fp8.py 2>/dev/null > $O/fp8.txt
find $O -name "*stats.csv" | head; rm -f $O/inf/*kernel_trace.csv $O/train/*kernel_trace.csv
tail -c 600 $O/bench_full.json | head -c 300; echo; cat $O/train_bench.json | cut -c1-200; cat $O/train_sdxl.json | cut -c1-200; cat $O/train_sdxl_fp8.json | cut -c1-200; cat $O/bench_b1.json | cut -c1-250
