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
