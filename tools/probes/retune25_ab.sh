# Race tile 25 (persistent ping-pong, csrc/gemm_ppp.hip) against every shape's incumbent inside the recorded B = 8 call, then A/B the two tune tables on the
# same box, alternating (old, new, old, new).  Writes gpurun_out/r06_retune25_*.json + the new table.
set -x
F="--no-train --no-cpu-baseline --no-single-view"
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_old.json
python bench.py --steps 10 --warmup 3 $F > gpurun_out/r06_retune25_old1.json 2> gpurun_out/r06_retune25_old1.err
GN_RETUNE=25 python bench.py --steps 2 --warmup 1 $F > gpurun_out/r06_retune25_race.json 2> gpurun_out/r06_retune25_race.err
cp genima_amd/gemm_tune_gfx950.json /tmp/tune_new.json
python bench.py --steps 10 --warmup 3 $F > gpurun_out/r06_retune25_new1.json 2> gpurun_out/r06_retune25_new1.err
cp /tmp/tune_old.json genima_amd/gemm_tune_gfx950.json
python bench.py --steps 10 --warmup 3 $F > gpurun_out/r06_retune25_old2.json 2> gpurun_out/r06_retune25_old2.err
cp /tmp/tune_new.json genima_amd/gemm_tune_gfx950.json
python bench.py --steps 10 --warmup 3 $F > gpurun_out/r06_retune25_new2.json 2> gpurun_out/r06_retune25_new2.err
cp /tmp/tune_new.json gpurun_out/gemm_tune_gfx950.json
python - <<'P'
import json
old, new = json.load(open("/tmp/tune_old.json")), json.load(open("/tmp/tune_new.json"))
ch = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(len(ch), "table entries changed:")
for k, (a, b) in sorted(ch.items()): print("  ", k, a, "->", b)
for n in ("old1", "new1", "old2", "new2"):
    try:
        j = json.loads(open(f"gpurun_out/r06_retune25_{n}.json").read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(j["ms_per_step"], 2), "median call", round(j.get("ms_per_call_median", 0), 2), "b2b value", round(j.get("value_back_to_back_no_d2h", 0), 1), "tiled_b1", j.get("tiled_b1", {}).get("ms_per_call"))
    except Exception as e: print(n, "ERR", e)
P
