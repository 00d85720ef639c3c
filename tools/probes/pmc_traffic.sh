# HBM traffic per kernel family of the tiled B=8 call: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined
# with the trace domains gpurun refuses), aggregated to gpurun_out/pmc_traffic/traffic.json in the format bench.py reads
# (sum of the counter over the launches of a family, in KB; FETCH_SIZE is to be doubled on gfx950 -- MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_traffic; mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-roofline --no-act --no-train --no-single-view > $O/$c.log 2>&1 || echo "pass $c failed"
done
python - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_traffic"
fams = [("gemm", ("gemm_dma_kernel", "gemm_pp_kernel", "gemm_s3_kernel", "gemm_kernel", "gemm_fp8_kernel", "splitk_reduce")), ("attn", ("attn_fwd",)), ("layernorm", ("layernorm_kernel",)),
        ("gn_stats", ("gn_stats_kernel",)), ("gn_apply", ("gn_apply_kernel",)), ("gn_fused", ("gn_fused_kernel",)), ("gn_finalize", ("gn_finalize_kernel",))]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = {f: {"sum_counter": 0.0, "launches": 0} for f, _ in fams}
    for path in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != c:
                continue
            for f, keys in fams:
                if any(k in r["Kernel_Name"] for k in keys):
                    agg[f]["sum_counter"] += float(r["Counter_Value"])
                    agg[f]["launches"] += 1
                    break
    out[c] = agg
import hashlib
root = os.environ.get("GRAFT_REPO_ROOT", ".")
# which code the counters belong to: the commit the caller passes in (the GPU box has no .git) and the hash of the library that ran
out["stamp"] = {"commit": os.environ.get("GIT_COMMIT", "unknown"),
                "lib_sha16": hashlib.sha256(open(root + "/genima_amd/libgenima_hip.so", "rb").read()).hexdigest()[:16],
                "command": "python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-roofline --no-act --no-train --no-single-view"}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
g = out["FETCH_SIZE"]["gemm"], out["WRITE_SIZE"]["gemm"]
print("gemm family: launches", g[0]["launches"], "bytes per launch", (2 * g[0]["sum_counter"] / max(1, g[0]["launches"]) + g[1]["sum_counter"] / max(1, g[1]["launches"])) * 1024)
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
