# kernel-trace window of the single-view B = 1 call (configs[1]): busy fraction, per-kernel time, gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_b1; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o b1 -- python bench.py --workload single_b1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train --no-single-view --no-act > $O/inf.log 2>&1
python tools/probes/trace_window.py $O/inf 100 4 > $O/b1_window.txt
head -60 $O/b1_window.txt
rm -f $O/inf/*kernel_trace.csv $O/inf/*/*kernel_trace.csv
