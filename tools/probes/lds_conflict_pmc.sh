# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel of a python script:  SCRIPT=tools/probes/attn_bwd_bench.py bash tools/probes/lds_conflict_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds_conflict; rm -rf $O; mkdir -p $O; cd $R
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p -o p --output-format csv -- python $SCRIPT > $O/run.log 2>&1 || echo failed
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/lds_conflict"
for f in sorted(glob.glob(O+"/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]; k=k[k.find("::")+2:][:70] if "::" in k else k[:70]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    rows=[]
    for k,v in acc.items():
        a=v.get("SQ_LDS_IDX_ACTIVE",0.0)
        if a<=0: continue
        rows.append((a, k, v))
    for a,k,v in sorted(rows, key=lambda r: -r[2].get('GRBM_GUI_ACTIVE',0))[:45]:
        c=v.get("SQ_LDS_BANK_CONFLICT",0.0); l=n[(k,"SQ_LDS_IDX_ACTIVE")]
        print(f"{k:72s} launches {l:5d}  LDS active {a/l:12.0f}  conflict {c/max(a,1)*100:5.1f} %  LDS/MFMA {v.get('SQ_INSTS_LDS',0)/max(v.get('SQ_INSTS_MFMA',1),1):5.2f}  LDS-active/(4 x MFMA-busy) {a/256/max(v.get('SQ_VALU_MFMA_BUSY_CYCLES',1)/1024,1):5.2f}  pipe busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(v.get('GRBM_GUI_ACTIVE',1)/8,1)*100:5.1f} %  cycles/launch {v.get('GRBM_GUI_ACTIVE',0)/8/l:10.0f}  total Mcyc {v.get('GRBM_GUI_ACTIVE',0)/8/1e6:8.2f}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
