# HBM traffic per kernel family of the tiled B=8 call: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined
# with the trace domains gpurun refuses), aggregated to gpurun_out/pmc_traffic/traffic.json in the format bench.py reads
# (sum of the counter over the launches of a family, in KB; FETCH_SIZE is to be doubled on gfx950 -- MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_traffic; mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-act --no-train --no-single-view > $O/$c.log 2>&1 || echo "pass $c failed"
done
python - <<'PY'
import csv, glob, hashlib, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_traffic"
STEPS, MARKER = 3, "image_f16_to_u8_kernel"  # the marker kernel closes a pipeline call: exactly one launch per call
fams = [("gemm", ("gemm_dma_kernel", "gemm_pp_kernel", "gemm_ppp_kernel", "gemm_s3_kernel", "gemm_kernel", "gemm_fp8_kernel", "splitk_reduce", "tblock_kernel", "conv3x3_gn_")),
        ("attn", ("attn_fwd",)), ("layernorm", ("layernorm_kernel",)), ("gn_stats", ("gn_stats_kernel",)), ("gn_apply", ("gn_apply_kernel",)),
        ("gn_fused", ("gn_fused_kernel",)), ("gn_finalize", ("gn_finalize_kernel",))]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for path in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    # the TIMED calls only: dispatches after the marker that closes the warm-up call (recording / autotune launches, the ACT program and the
    # warm-up call are cut off), up to the last marker -- both passes then count the same STEPS x launches_per_call population
    marks = [i for i, r in enumerate(rows) if MARKER in r[1]]
    assert len(marks) > STEPS, f"{c}: only {len(marks)} {MARKER} launches"
    rows = rows[marks[-STEPS - 1] + 1:marks[-1] + 1]
    agg = {f: {"sum_counter": 0.0, "launches": 0, "ops": 0} for f, _ in fams}
    for _, name, val in rows:
        for f, keys in fams:
            if any(k in name for k in keys):
                agg[f]["sum_counter"] += val
                agg[f]["launches"] += 1
                agg[f]["ops"] += 0 if "splitk_reduce" in name else 1  # a split-K gn_gemm is ONE op of the program: its reduce launch rides along
                break
    agg["_all"] = {"sum_counter": sum(r[2] for r in rows), "launches": len(rows)}
    out[c] = agg
root = os.environ.get("GRAFT_REPO_ROOT", ".")
# which code the counters belong to: the commit the caller passes in (the GPU box has no .git), the hash of the library that ran, and the
# number of GEMM-family launches per call the two passes saw (bench.py refuses a file whose count disagrees with its own program)
g = out["FETCH_SIZE"]["gemm"], out["WRITE_SIZE"]["gemm"]
assert g[0]["launches"] == g[1]["launches"], (g[0]["launches"], g[1]["launches"])
out["stamp"] = {"commit": os.environ.get("GIT_COMMIT", "unknown"),
                "lib_sha16": hashlib.sha256(open(root + "/genima_amd/libgenima_hip.so", "rb").read()).hexdigest()[:16],
                "src_sha16": __import__("genima_amd.build", fromlist=["source_sha16"]).source_sha16(),
                "calls": STEPS, "gemm_launches_per_call": g[0]["launches"] / STEPS, "gemm_ops_per_call": g[0]["ops"] / STEPS,
                "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-act --no-train --no-single-view; dispatches of the 3 timed calls (cut at image_f16_to_u8_kernel)"}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
print("gemm family: launches per call", g[0]["launches"] / STEPS, "ops per call", g[0]["ops"] / STEPS, "bytes per op", (2 * g[0]["sum_counter"] + g[1]["sum_counter"]) / max(1, g[0]["ops"]) * 1024)
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
