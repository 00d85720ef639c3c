"""Aggregate a rocprofv3 kernel trace over its LAST `ms` milliseconds (the timed steps of a bench run): per-kernel time, launches, and
how busy the GPU was (sum of kernel durations / window; < 1 means launch gaps: the host could not keep the queue full).
usage: trace_window.py <dir> <window ms> [steps] [marker]
With a `marker` (a kernel that runs exactly ONCE per step: `image_f16_to_u8_kernel` closes an inference call, `adamw_kernel` a train step)
the window is cut to EXACTLY `steps` steps -- from the end of the marker launch `steps` + 1 from the end to the end of the last one -- instead
of a fixed number of milliseconds (round 2's windows held ~2.85 calls and were divided by 3)."""
import csv, glob, sys, collections, re
d, win = sys.argv[1], float(sys.argv[2]) * 1e6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t1 = max(r[1] for r in rows); t0 = t1 - win
marker = sys.argv[4] if len(sys.argv) > 4 else None
if marker:
    ends = sorted(e for s, e, n in rows if marker in n)
    assert len(ends) > steps, f"only {len(ends)} launches of {marker}"
    t0, t1 = ends[-steps - 1], ends[-1]
    win = float(t1 - t0)
    rows = [r for r in rows if r[1] <= t1]
sel = [r for r in rows if r[0] >= t0]
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for s, e, n in sel:
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    agg[n][0] += 1; agg[n][1] += (e - s)
# union of intervals (streams overlap)
cur_s, cur_e = None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(v[1] for v in agg.values())
print(f"window {win/1e6:.1f} ms, {len(sel)} launches ({len(sel)/steps:.0f}/step), kernel time sum {tot/1e6:.1f} ms, GPU busy (union) {busy/1e6:.1f} ms = {busy/win:.1%}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t/1e6/steps:8.3f} ms/step {c/steps:7.1f} launches/step {t/c/1e3:8.1f} us  {n[:110]}")
# idle gaps between consecutive kernels (by end -> next start), aggregated by the kernel that FOLLOWS the gap
gaps = collections.defaultdict(lambda: [0, 0.0]); hist = collections.Counter()
pe, pn = None, None
for s, e, n in sel:
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    if pe is not None and s > pe:
        g = s - pe
        gaps[(pn[:40], n[:40])][0] += 1; gaps[(pn[:40], n[:40])][1] += g
        hist[min(9, int(g // 5000))] += g
    if pe is None or e > pe: pe, pn = e, n
print("idle by gap length (5 us bins, last = >= 45 us), ms/step:", {k * 5: round(v / 1e6 / steps, 2) for k, v in sorted(hist.items())})
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6/steps:7.3f} ms/step idle  {c/steps:6.1f} gaps/step  avg {t/c/1e3:7.1f} us   {a} -> {b}")
# CONTEXT=<kernel substring>: the launches around the last occurrence of that kernel (start / end relative to it, queue) -- what the GPU was
# doing across one particular seam
import os
ctx = os.environ.get("CONTEXT")
if ctx:
    full = []
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            full.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    full.sort()
    idx = max(i for i, r in enumerate(full) if ctx in r[3])
    base = full[idx][0]
    for s, e, q, n in full[max(0, idx - 14):idx + 6]:
        n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
        print(f"  start {(s - base) / 1e3:9.1f} us  end {(e - base) / 1e3:9.1f} us  queue {q:>3}  {n[:90]}")
