"""One B = 8 call as sub-batches in flight: does running the eight episodes of a call as 2 x B = 4 (or 4 x B = 2) recorded programs on their own
streams, joined at the end of the call, beat the one B = 8 program?  (bench.py's `two_calls_in_flight` says two B = 8 calls in flight take 86 ms
per call against 95: concurrency fills what one program leaves idle -- but a half batch's launches are also less efficient.)

  python tools/probes/half_batches.py [--calls 6]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def build(pipe, dev, B, n, steps):
    progs = []
    for _ in range(n):
        pipe._progs.clear()
        progs.append(pipe.program(B, 512, 512, steps))
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    ids, img, lat = bench.synthetic_inputs(pipe, B, 512, 512, dev, 0)
    for io, st in zip(progs, streams):
        io.engine.use_stream(st)
        io.ids.copy_(ids.to(torch.int32)); io.image_u8.copy_(img); io.noise.copy_(lat.permute(0, 2, 3, 1))
    torch.cuda.synchronize(dev)
    return progs, streams


def timed(progs, streams, dev, calls, join, graph=False):
    def one():
        for io in progs:
            io.engine.launch() if graph else io.engine.run()
        if join and len(progs) > 1:  # the call ends when every sub-batch has: nobody starts the next call before that
            evs = []
            for st in streams:
                e = torch.cuda.Event()
                e.record(st)
                evs.append(e)
            for st in streams:
                for e in evs:
                    st.wait_event(e)
    for _ in range(2):
        one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(calls):
        one()
    torch.cuda.synchronize(dev)
    return 1000.0 * (time.perf_counter() - t0) / calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=6)
    ap.add_argument("--graph", action="store_true", help="replay each program as one captured hipGraph")
    ap.add_argument("--configs", default="8x1,4x1,4x2,2x1,2x4,8x2", help="B x programs, comma separated")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from genima_amd import configs
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family("sd-turbo"), seed=0, gen_device=dev)
    pipe.to(dev)
    for m in (pipe.vae, pipe.text_encoder, pipe.unet, pipe.controlnet):
        m._sd = None
    torch.cuda.empty_cache()
    saved = dict(pipe._progs)
    rows = []
    for B, n in [tuple(int(v) for v in c.split("x")) for c in args.configs.split(",")]:
        progs, streams = build(pipe, dev, B, n, 5)
        if args.graph:
            for io, st in zip(progs, streams):
                with torch.cuda.stream(st):
                    io.engine.run()
                    st.synchronize()
                    io.engine.capture()
        for rep in range(2):
            j = timed(progs, streams, dev, args.calls, True, args.graph)
            f = timed(progs, streams, dev, args.calls, False, args.graph) if n > 1 else float("nan")
            rows.append((B, n, rep, j, f))
            print(f"{'graph ' if args.graph else ''}{n} x B = {B}: joined {j:8.2f} ms per round of {n * B} episodes ({j / (n * B):6.2f} ms / episode)   free-running {f:8.2f}", flush=True)
        del progs, streams
        torch.cuda.empty_cache()
    pipe._progs.clear()
    pipe._progs.update(saved)


if __name__ == "__main__":
    main()
