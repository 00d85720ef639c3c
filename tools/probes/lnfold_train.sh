cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
GN_LN_FOLD=$m rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lnft$m -o t -- python $R/bench_train.py --steps 5 --warmup 3 > $R/gpurun_out/lnft$m.json 2> $R/gpurun_out/lnft$m.err
tail -1 $R/gpurun_out/lnft$m.json | cut -c1-200
done
ls $R/gpurun_out/lnft0
