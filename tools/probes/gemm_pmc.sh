cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gemm_pmc; rm -rf $O; mkdir -p $O; cd $R
run() { # tag, env...
  tag=$1; shift
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" ; do
    env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/$tag -o p --output-format csv -- python tools/probes/gemm_one.py > $O/$tag.log 2>&1 || echo "pass failed: $set"
    python - "$O/$tag" "$tag" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = 0
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); 
disp = set()
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "gemm" in r["Kernel_Name"]]
    n = len(rows); dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / max(1, n)
    print(sys.argv[2], "launches", n, "avg ns", dur)
print(sys.argv[2], {k: v / max(1, n) for k, v in agg.items()})
PY
    rm -rf $O/$tag
  done
}
run lin8192x640x640_t9 KIND=linear M=8192 N=640 K=640 CFG=9
run lin2048x1280x1280_t9 KIND=linear M=2048 N=1280 K=1280 CFG=9
run conv320_64_t12 KIND=conv CIN=320 COUT=320 HW=64 CFG=12
run conv512_128_pp KIND=conv CIN=512 COUT=512 HW=128 CFG=14
