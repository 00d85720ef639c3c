"""Probe: two B = 8 tiled-call programs (own buffers) on two HIP streams, the second started half a call after the first, both looping
back to back -- one call's VAE decode then runs beside the other's denoise steps.  Throughput against the single program."""
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
ids8, img8, lat8 = bench.synthetic_inputs(pipe, 8, H, W, dev, 0)
progs = []
for r in range(2):
    pipe._progs.clear()
    progs.append(pipe.program(8, H, W, steps))
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for io, s in zip(progs, streams):
    io.engine.use_stream(s)
    io.ids.copy_(ids8.to(torch.int32)); io.image_u8.copy_(img8); io.noise.copy_(lat8.permute(0, 2, 3, 1))
torch.cuda.synchronize()
N = 12
for _ in range(3):
    progs[0].engine.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2 * N):
    progs[0].engine.run()
torch.cuda.synchronize()
single = (time.perf_counter() - t0) / (2 * N) * 1e3
print(f"one program, calls back to back: {single:.2f} ms per call")
for delay in (0.0, 0.05):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    progs[0].engine.run()
    time.sleep(delay)
    for i in range(N - 1):
        progs[1].engine.run()
        progs[0].engine.run()
    progs[1].engine.run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0 - 0.0) / (2 * N) * 1e3
    print(f"two programs on two streams, second started {delay * 1e3:.0f} ms late: {t:.2f} ms per call ({single / t:.3f}x)")
