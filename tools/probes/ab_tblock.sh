#!/bin/bash
# Same-box A/B of the fused transformer-block chains (GN_TBLOCK=1/0) on the driver's headline workload; prints the two JSON lines' key figures.
#   tools/probes/ab_tblock.sh [steps]     (on the GPU box; writes gpurun_out/ab_tblock_{1,0}.json)
steps=${1:-10}
mkdir -p gpurun_out
for v in ${AB_ORDER:-1 0 1 0}; do
  GN_TBLOCK=${AB_TBLOCK:-$v} GN_CONV_GN=${AB_CONV_GN:-$v} GN_TBLOCK_FRONT=${AB_FRONT:-$v} python bench.py --steps $steps --warmup 3 --no-train --no-single-view 2>/dev/null | tail -1 > gpurun_out/ab_tblock_$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_tblock_$v.json"))
rows = [(r["kernel"], round(r["ms_per_call"], 2), r["launches"], round(r.get("achieved", 0), 1)) for r in d.get("roofline_extra", [])[:9]]
print("GN_TBLOCK=GN_CONV_GN=$v", round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms  gemm family", round(d["roofline"]["achieved"], 1), "TF/s", rows, flush=True)
PY
done
