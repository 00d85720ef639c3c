"""Compare two --dump-ops tables (bench.py) by kind and by (kind, shape):  ops_diff.py A.csv B.csv [rows]"""
import collections, csv, sys
a, b = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
t = {}
for name in (a, b):
    t[name] = {(r["kind"], r["shape"]): (int(r["launches"]), float(r["total_ms"]), float(r["avg_us"])) for r in csv.DictReader(open(name))}
kinds = {}
for name in (a, b):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for (k, _), v in t[name].items():
        agg[k][0] += v[0]; agg[k][1] += v[1]
    kinds[name] = agg
print(f"A = {a}\nB = {b}")
for k in sorted(set(kinds[a]) | set(kinds[b]), key=lambda k: -kinds[b].get(k, [0, 0])[1]):
    x, y = kinds[a].get(k, [0, 0.0]), kinds[b].get(k, [0, 0.0])
    print(f"  {k:18s} A {x[0]:5d} {x[1]:8.3f} ms | B {y[0]:5d} {y[1]:8.3f} ms | A-B {x[1] - y[1]:+7.3f}")
print(f"  {'total':18s} A {sum(v[0] for v in kinds[a].values()):5d} {sum(v[1] for v in kinds[a].values()):8.3f} ms | B {sum(v[0] for v in kinds[b].values()):5d} {sum(v[1] for v in kinds[b].values()):8.3f} ms")
keys = sorted(set(t[a]) | set(t[b]), key=lambda k: -abs(t[a].get(k, (0, 0, 0))[1] - t[b].get(k, (0, 0, 0))[1]))
for k in keys[:top]:
    x, y = t[a].get(k, (0, 0, 0)), t[b].get(k, (0, 0, 0))
    print(f"  {k[0]:10s} {k[1]:26s} A {x[0]:4d} x {x[2]:7.1f} us = {x[1]:7.3f} | B {y[0]:4d} x {y[2]:7.1f} us = {y[1]:7.3f} | {x[1] - y[1]:+.3f}")
