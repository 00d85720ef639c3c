# HBM traffic per (kernel, grid) of the tiled B=8 call: which launches move more bytes than their algorithm needs.
# Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with the kernel trace; cut to the 3 timed calls at the marker kernel;
# output gpurun_out/pmc_by_grid/by_grid.csv: kernel, grid, wg, launches per call, avg us, FETCH MB (x2-corrected) and WRITE MB per launch, TB/s
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_by_grid; mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-act --no-train --no-single-view ${BENCH_EXTRA} > $O/$c.log 2>&1 || echo "pass $c failed"
done
python - <<'PY'
import csv, glob, os, re, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_by_grid"
STEPS, MARKER = 3, "image_f16_to_u8_kernel"
def clean(n):
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for path in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""),
                             r.get("LDS_Block_Size", ""), int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if MARKER in r[1]]
    rows = rows[marks[-STEPS - 1] + 1:marks[-1] + 1]
    for _, name, val, grid, wg, lds, dur in rows:
        a = agg[(clean(name), grid, wg)]
        a[c] += val
        if c == "FETCH_SIZE": a["n"] += 1; a["ns"] += dur
with open(O + "/by_grid.csv", "w") as f:
    f.write("kernel,grid,wg,launches_per_call,avg_us,fetch_MB_x2_per_launch,write_MB_per_launch,TBps,total_GB_per_call\n")
    for (k, grid, wg), a in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        n = max(1, a["n"]); fm = 2 * a["FETCH_SIZE"] * 1024 / n / 1e6; wm = a["WRITE_SIZE"] * 1024 / n / 1e6; us = a["ns"] / n / 1e3
        f.write(f"{k},{grid},{wg},{a['n'] / STEPS:.1f},{us:.1f},{fm:.2f},{wm:.2f},{(fm + wm) / max(us, 1e-3):.2f},{(fm + wm) * a['n'] / STEPS / 1e3:.2f}\n")
print(open(O + "/by_grid.csv").read()[:6000])
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
