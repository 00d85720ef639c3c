"""Ping-pong 256x256 GEMM (csrc/gemm_pp.hip, tile 15) against the LDS-DMA 256x256 tile (7) and the tuned tile per shape:
bitwise equality of the outputs (K is walked identically by every tile), a race screen (repeat runs must be bit-identical) and
HIP-event timings on random data.  Run on the GPU box:  python tools/bench_gemm_pp.py > gpurun_out/gemm_pp.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
PP, D256 = 14, 6


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


def h(*shape, s=0.5):
    return (torch.randn(*shape, device="cuda") * s).half()


def run(name, fn, flops, ref=None):
    outs, times = {}, {}
    for tag, cfg in (("d256", D256), ("pp", PP), ("tuned", -1)):
        E.no_table = cfg >= 0
        E.lib.gn_set_gemm_tile_override(cfg)
        o = fn().clone()
        for _ in range(4):  # race screen
            o2 = fn()
            if not torch.equal(o, o2):
                print(f"!! {name} [{tag}]: repeat run differs (max {float((o.float() - o2.float()).abs().max()):.3e})", flush=True)
                break
        outs[tag] = o
        times[tag] = timeit(fn)
    E.lib.gn_set_gemm_tile_override(-1)
    E.no_table = False
    same = torch.equal(outs["pp"], outs["d256"])
    d = float((outs["pp"].float() - outs["d256"].float()).abs().max())
    extra = ""
    if ref is not None:
        r = ref()
        extra = f" relL2 vs torch {float((outs['pp'].float() - r).norm() / r.norm()):.2e}"
    tf = {k: flops / v / 1e9 for k, v in times.items()}
    print(f"{name:44s} pp {tf['pp']:7.1f} TF ({times['pp'] * 1e3:7.1f} us) | d256 {tf['d256']:7.1f} | tuned {tf['tuned']:7.1f} | "
          f"pp/tuned {tf['pp'] / tf['tuned']:.2f} | bitwise={'OK' if same else 'DIFF max %.3e' % d}{extra}", flush=True)


B = int(os.environ.get("B", "8"))
# correctness on small / ragged problems against torch (transpose-detecting: non-square, distinct M / N / K)
x, w, b = h(1000, 320), h(328, 320), h(328)
run("linear ragged 1000x328x320", lambda: E.linear(x, w, b), 2.0 * 1000 * 328 * 320, lambda: x.float() @ w.float().t() + b.float())
xc, wc, bc = h(2, 24, 40, 64), h(72, 9 * 64, s=0.05), h(72)


def conv_ref():
    wt = wc.float().view(72, 3, 3, 64).permute(0, 3, 1, 2)
    return torch.nn.functional.conv2d(xc.float().permute(0, 3, 1, 2), wt, bc.float(), padding=1).permute(0, 2, 3, 1)


run("conv3x3 64->72 @24x40 (ragged)", lambda: E.conv2d(xc, wc, bc), 2.0 * 2 * 24 * 40 * 72 * 576, conv_ref)
x2, w2 = h(2, 16, 16, 128), h(128, 9 * 192, s=0.05)
xb = h(2, 16, 16, 64)
run("conv3x3 concat 128+64->128 @16 ups", lambda: E.conv2d(x2, w2, None, x2=xb, upsample2x=True), 2.0 * 2 * 32 * 32 * 128 * 1728)
w3 = w2[:, :1152].contiguous()
run("conv3x3 128->128 stride 2", lambda: E.conv2d(x2, w3, None, stride=2), 2.0 * 2 * 8 * 8 * 128 * 1152)

# the hot shapes
for cin, cout, hw in [(320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32),
                      (1280, 1280, 16), (2560, 1280, 16), (1280, 1280, 8), (512, 512, 64), (512, 512, 128), (256, 256, 256),
                      (256, 256, 512), (128, 128, 512), (512, 512, 256)]:
    if B * hw * hw * max(cin, cout) * 2 > 3.5e9:
        continue
    x = h(B, hw, hw, cin)
    w = h(cout, 9 * cin, s=0.02)
    b = h(cout)
    run(f"conv3x3 {cin}->{cout} @{hw}x{hw}", lambda: E.conv2d(x, w, b), 2.0 * B * hw * hw * cout * 9 * cin)
for tok, k, n in [(4096, 320, 320), (4096, 320, 640), (4096, 320, 960), (4096, 1280, 320), (1024, 640, 640), (1024, 640, 1920),
                  (1024, 2560, 640), (256, 1280, 1280), (256, 1280, 3840), (256, 5120, 1280), (8192, 4096, 4096)]:
    x = h(B * tok, k)
    w = h(n, k, s=0.05)
    b = h(n)
    r = h(B * tok, n)
    run(f"linear {B}x{tok} K={k} N={n} (+res)", lambda: E.linear(x, w, b, residual=r), 2.0 * B * tok * k * n)
