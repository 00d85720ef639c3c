"""Probe: how well the two streams of the recorded B = 8 program are balanced.  Ops recorded between `fork` and `main` run on the side stream
(ControlNet), ops between `main` and `join` on the main stream (UNet encoder + mid block); the decoder starts at the join.  Per section:
sum of the per-op HIP-event times of one op-by-op replay (serialised, so the sums say how long each chain is, not how they overlap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genima_amd import configs
from genima_amd.pipeline import StableDiffusionControlNetPipeline
import bench

dev = torch.device("cuda", 0)
pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family("sd-turbo"), seed=0, gen_device=dev).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ids, img, lat = bench.synthetic_inputs(pipe, B, 512, 512, dev, 0)
pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=5, guidance_scale=0.0, output_type="pt")
io = pipe.program(B, 512, 512, 5)
E = io.engine
n = E.num_ops
evs = [E.event() for _ in range(n + 1)]
E.synchronize(); E.event_record(evs[0])
for i in range(n):
    E.run(i, i + 1); E.event_record(evs[i + 1])
E.synchronize()
ms = [E.event_elapsed_ms(evs[i], evs[i + 1]) for i in range(n)]
# stream markers in meta order: fork, main, join (kind == "stream"); which is which follows from their cyclic order
state, sec, out = "main0", 0.0, []
k = 0
for i, m in enumerate(E.meta[:n]):
    if m["kind"] == "stream":
        out.append((state, sec)); sec = 0.0
        state = ("side", "main", "after")[k % 3]; k += 1
    else:
        sec += ms[i]
out.append((state, sec))
for i in range(0, len(out)):
    print(f"{out[i][0]:6s} {out[i][1]:8.3f} ms")
