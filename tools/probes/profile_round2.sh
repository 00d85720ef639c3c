# Round-2 evidence run (on the GPU box): the default bench line, rocprofv3 kernel stats of the same commands, PMC traffic passes,
# the GEMM power / DMA probes behind DESIGN.md section 3, and the micro-benchmarks.  Summaries are copied to profiles/r02_* by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02; mkdir -p $O
cd $R
python bench.py > $O/bench_full.json 2> $O/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o b8 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-single-view > $O/inf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t8 -- python bench_train.py --steps 5 --warmup 2 > $O/train.log 2>&1
python bench_train.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/train_bench.json
python bench_train.py --family sdxl-turbo 2>/dev/null | tail -1 > $O/train_sdxl.json
python bench_train.py --family sdxl-turbo --fp8 2>/dev/null | tail -1 > $O/train_sdxl_fp8.json
python bench.py --dump-ops $O/ops_b8.csv --no-cpu-baseline --no-train --no-single-view > /dev/null 2>&1
python tools/bench_attn.py 2>/dev/null > $O/attn.txt
python tools/bench_gemm_pp.py > $O/gemm_pp.txt 2>/dev/null
python tools/probes/gemm_data_dep.py 2>/dev/null | grep cfg > $O/gemm_data_dep.txt
tools/probes/dma_bw > $O/dma_bw.txt 2>&1
for a in 0 1 2 3 4 5; do GN_PP_ABL=$a python tools/probes/gemm_pp_abl.py 2>/dev/null | grep ABL; done > $O/gemm_pp_abl.txt
find $O -name "*stats.csv" | head; rm -f $O/inf/*kernel_trace.csv $O/train/*kernel_trace.csv $O/inf/*/*kernel_trace.csv $O/train/*/*kernel_trace.csv
tail -c 400 $O/bench_full.json | head -c 300; echo; cut -c1-160 $O/train_bench.json; cut -c1-160 $O/train_sdxl.json; cut -c1-160 $O/train_sdxl_fp8.json
