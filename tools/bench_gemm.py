"""Micro-benchmark of the MFMA implicit-GEMM kernel over the hot-path shapes x tile configurations (run on the GPU box):
    python tools/bench_gemm.py > gpurun_out/gemm_tiles.txt
Prints achieved TFLOP/s per (shape, tile cfg); used to set the efficiency table / heuristic in csrc/gemm.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.no_table = True  # time the configuration forced through gn_set_gemm_tile_override, not the tuned one
CFG = ["256x128", "128x128", "128x64", "64x64", "256x64", "128x256",
       "D256x256", "D256x128", "D128x128", "D128x64", "D64x64", "D256x64", "D128x320", "D256x320", "PP256x256",
       "S3_128x128", "S3_128x64", "S3_64x64", "S3_256x64", "S3_128x160", "S3_64x160", "S3_64x320", "D128x160", "D128x320w8"]
ALL = tuple(int(c) for c in os.environ.get("CFGS", "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18").split(","))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


def run(name, fn, flops, cfgs=ALL):
    row = []
    for c in cfgs:
        E.lib.gn_set_gemm_tile_override(c)
        try:
            ms = timeit(fn)
            row.append(f"{CFG[c]}: {flops / ms / 1e9:7.1f} TF ({ms * 1000:7.1f} us)")
        except Exception as e:  # noqa: BLE001
            row.append(f"{CFG[c]}: ERR {str(e)[:40]}")
    E.lib.gn_set_gemm_tile_override(-1)
    ms = timeit(fn)
    print(f"{name:46s} | " + " | ".join(row) + f" | auto: {flops / ms / 1e9:7.1f} TF", flush=True)


def h(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).half()


B = int(os.environ.get("B", "8"))
print(f"# batch {B}")
# conv 3x3 (Cin, Cout, H) at the UNet / VAE levels
for cin, cout, hw in [(320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32),
                      (1280, 1280, 16), (2560, 1280, 16), (1280, 1280, 8), (2560, 1280, 8), (512, 512, 64), (512, 512, 128),
                      (256, 256, 256), (128, 128, 512)]:
    x = h(B, hw, hw, cin)
    w = h(cout, 9 * cin)
    b = h(cout)
    run(f"conv3x3 {cin}->{cout} @{hw}x{hw}", lambda: E.conv2d(x, w, b), 2.0 * B * hw * hw * cout * 9 * cin)
# linears (tokens, K, N)
for tok, k, n in [(4096, 320, 320), (4096, 320, 640), (4096, 1280, 320), (1024, 640, 640), (1024, 640, 1280), (1024, 2560, 640),
                  (256, 1280, 1280), (256, 1280, 2560), (256, 5120, 1280), (64, 1280, 1280)]:
    x = h(B * tok, k)
    w = h(n, k)
    b = h(n)
    run(f"linear {B}x{tok} K={k} N={n}", lambda: E.linear(x, w, b), 2.0 * B * tok * k * n)
for tok, c in [(4096, 320), (1024, 640), (256, 1280)]:
    x = h(B * tok, c)
    w = h(8 * c, c)
    b = h(8 * c)
    run(f"geglu {B}x{tok} C={c}", lambda: E.linear(x, w, b, act=5), 2.0 * B * tok * c * 8 * c, cfgs=tuple(c for c in ALL if c in (0, 1, 4, 5, 6, 7, 8, 11, 15, 18)))
