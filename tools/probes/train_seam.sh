# What runs on the GPU around the backward -> optimizer seam of the ControlNet fine-tune step (rocprofv3 kernel trace of bench_train.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train_seam; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t8 -- python bench_train.py --steps 4 --warmup 2 > $O/train.log 2>&1
CONTEXT=sumsq_partial python tools/probes/trace_window.py $O/t 360 3 adamw_kernel | tail -22 > $O/seam.txt
rm -rf $O/t
cat $O/seam.txt
