"""Probe: the B = 8 tiled call as TWO concurrent half-batch programs (B = 4 each) on separate HIP streams vs the single B = 8 program.
T(B) = 24 + 10.4 B ms on MI355X (33.8 / 44.8 / 66.1 / 107.0 ms at B = 1 / 2 / 4 / 8): 24 ms of every call is per-launch latency that an
independent chain could fill."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genima_amd import configs
from genima_amd.pipeline import StableDiffusionControlNetPipeline
import bench

dev = torch.device("cuda", 0)
pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family("sd-turbo"), seed=0, gen_device=dev).to(dev)
for m in (pipe.vae, pipe.text_encoder, pipe.unet, pipe.controlnet):
    m._sd = None
H = W = 512
steps = 5


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ids8, img8, lat8 = bench.synthetic_inputs(pipe, 8, H, W, dev, 0)
t8 = timeit(lambda: pipe(prompt_ids=ids8, image=img8, latents=lat8, num_inference_steps=steps, guidance_scale=0.0, output_type="pt"))
print(f"single B=8 program: {t8:.2f} ms")

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Bp = 8 // parts
progs = []
for r in range(parts):
    pipe._progs.clear()  # force a distinct program (own buffers) per replica
    io = pipe.program(Bp, H, W, steps)
    progs.append(io)
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
for io, s in zip(progs, streams):
    io.engine.use_stream(s)
    io.ids.copy_(ids8[:Bp].to(torch.int32)); io.image_u8.copy_(img8[:Bp]); io.noise.copy_(lat8[:Bp].permute(0, 2, 3, 1))
torch.cuda.synchronize()


def run_split():
    for io in progs:
        io.engine.run()


t = timeit(run_split)
print(f"{parts} concurrent B={Bp} programs: {t:.2f} ms  ({t8 / t:.3f}x)")
one = timeit(lambda: progs[0].engine.run())
print(f"one B={Bp} program alone: {one:.2f} ms")
