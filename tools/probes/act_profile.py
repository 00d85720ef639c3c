"""The ACT controller forward alone (B = 8 tiled 512 x 512 frames -> [8, 20, 8] actions): per-op HIP-event table of its recorded program."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd import configs
from genima_amd.act import GenimaACT
dev = torch.device("cuda:0")
fam = configs.family("sd-turbo")
agent = GenimaACT(fam["act"], None, fam["act_text"], None, device=dev, seed=0)
B = 8
tiled = torch.randint(0, 255, (B, 512, 512, 3), dtype=torch.uint8, device=dev)
state = torch.zeros(B, 8, device=dev)
tokens = torch.zeros(B, 77, dtype=torch.int32, device=dev); tokens[:, 0] = 49406; tokens[:, 1:6] = 320; tokens[:, 6] = 49407
for _ in range(3): agent.act_tiled(tiled, state, tokens)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): agent.act_tiled(tiled, state, tokens)
b.record(); torch.cuda.synchronize()
print(f"act_tiled: {a.elapsed_time(b) / 20:.3f} ms per call")
# per-op table of the recorded ACT program (HIP events around every op of one replay)
import collections
io = agent._run(torch.stack([tiled[:, y:y + 256, x:x + 256] for (x, y) in ((0, 0), (256, 0), (0, 256), (256, 256))], dim=1), state, tokens)
E = io.engine
n = E.num_ops
evs = [E.event() for _ in range(n + 1)]
E.synchronize(); E.event_record(evs[0])
for i in range(n):
    E.run(i, i + 1); E.event_record(evs[i + 1])
E.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for i, m in enumerate(E.meta[:n]):
    k = (m["kind"], tuple(m["shape"]))
    agg[k][0] += 1; agg[k][1] += E.event_elapsed_ms(evs[i], evs[i + 1])
print(f"{n} ops, {sum(v[1] for v in agg.values()):.3f} ms op by op")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:12s} {'x'.join(map(str, k[1])):24s} {v[0]:4d} launches {v[1]*1e3:8.1f} us  ({v[1]*1e3/v[0]:.1f} each)")
