# GPU busy share and the launch gaps of the B = 1 calls (single view, tiled): rocprofv3 kernel trace of `bench.py --workload ...`, exact-N-call window
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1win; rm -rf $O; mkdir -p $O
cd $R
for w in single_b1 tiled_b1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o k -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-single-view > $O/$w.log 2>&1
python tools/probes/trace_window.py $O/$w 40 6 image_f16_to_u8_kernel > $O/${w}_window.txt
rm -f $O/$w/*kernel_trace.csv $O/$w/*/*kernel_trace.csv
done
head -3 $O/single_b1_window.txt
