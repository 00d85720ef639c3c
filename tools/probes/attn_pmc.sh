cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_pmc
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python tools/probes/attn_pmc.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
find $O -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/attn_pmc"
for f in sorted(glob.glob(O+"/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        n[(k,r["Counter_Name"])]+=1
    print(f)
    for k,v in acc.items():
        if "attn" not in k: continue
        for c,x in v.items(): print("  ",k[:40],c,x/n[(k,c)])
PY
