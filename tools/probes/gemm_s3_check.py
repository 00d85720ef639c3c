"""3-stage ring tiles (gemm_s3.hip, cfg 15..21; cfg 22 = 2-stage 128x160) against the 2-stage tiles: bitwise equality, race screen, timing on the odd shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
cases = []
x, w, b = h(8, 16, 16, 2560), h(1280, 9 * 2560, sc=0.02), h(1280)
cases.append(("conv 2560->1280@16", lambda: E.conv2d(x, w, b)))
x2, w2, b2 = h(1000, 328), h(72, 328), h(72)
cases.append(("linear ragged 1000x72x328", lambda: E.linear(x2, w2, b2)))
x3, w3 = h(2, 24, 40, 64), h(72, 9 * 64, sc=0.05)
cases.append(("conv ragged 64->72", lambda: E.conv2d(x3, w3, None, stride=2)))
x4, x4b, w4 = h(2, 16, 16, 128), h(2, 16, 16, 64), h(128, 9 * 192, sc=0.05)
cases.append(("conv concat ups", lambda: E.conv2d(x4, w4, None, x2=x4b, upsample2x=True)))
x5, w5, r5 = h(8192, 640), h(640, 640, sc=0.05), h(8192, 640)
cases.append(("linear 8192x640x640 +res", lambda: E.linear(x5, w5, None, residual=r5)))
x6, w6, b6, r6 = h(2048, 1280), h(1280, 1280, sc=0.03), h(1280), h(2048, 1280)
cases.append(("linear 2048x1280x1280 +res", lambda: E.linear(x6, w6, b6, residual=r6)))
x7, w7, b7, r7 = h(32768, 320), h(320, 320, sc=0.05), h(320), h(32768, 320)
cases.append(("linear 32768x320x320 +res", lambda: E.linear(x7, w7, b7, residual=r7)))
x8, w8, b8 = h(8, 32, 32, 640), h(640, 9 * 640, sc=0.02), h(640)
cases.append(("conv 640->640@32", lambda: E.conv2d(x8, w8, b8)))
x9, w9, b9 = h(8, 64, 64, 320), h(320, 9 * 320, sc=0.02), h(320)
cases.append(("conv 320->320@64", lambda: E.conv2d(x9, w9, b9)))
x10, w10, b10 = h(8, 16, 16, 1280), h(1280, 9 * 1280, sc=0.02), h(1280)
cases.append(("conv 1280->1280@16", lambda: E.conv2d(x10, w10, b10)))
for name, fn in cases:
    E.lib.gn_set_gemm_tile_override(9); ref = fn().clone()
    for cfg in (15, 16, 17, 18, 19, 20, 21, 22):
        E.lib.gn_set_gemm_tile_override(cfg)
        o = fn().clone(); ok = torch.equal(o, ref)
        rep = all(torch.equal(fn(), o) for _ in range(5))
        a, bb = E.event(), E.event(); E.event_record(a)
        for _ in range(10): fn()
        E.event_record(bb); ms = E.event_elapsed_ms(a, bb) / 10
        print(f"{name:28s} cfg {cfg}: bitwise {'OK' if ok else 'DIFF %.3e' % float((o.float()-ref.float()).abs().max())} repeat {'OK' if rep else 'RACE'} {ms*1e3:8.1f} us", flush=True)
