"""Time vs K at M = 32768, N = 320 (the 64x64-latent level's C -> C Linears: 175 launches per tiled call) for the tiles that fit N = 320 without
waste: slope = cost per 64-wide K tile (a latency-bound K loop shows ~ one HBM round trip per stage), intercept = launch + prologue + epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def t(fn, n=200):
    for _ in range(20): fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(n): fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / n * 1e3
M, N = 32768, 320
names = {22: "128x160 2-stage", 19: "128x160 ring", 20: "64x160 ring", 21: "64x320 ring", 12: "128x320", 13: "256x320", 9: "128x64", 16: "128x64 ring", 10: "64x64"}
for cfg in (22, 19, 20, 21, 12, 13, 9, 16, 10):
    E.lib.gn_set_gemm_tile_override(cfg)
    for res in (0, 1):
        row = []
        for K in (64, 128, 192, 320, 640, 1280):
            x, w, b, r = h(M, K), h(N, K, sc=0.05), h(N), h(M, N)
            row.append("K=%d %.1f" % (K, t(lambda: E.linear(x, w, b, residual=r if res else None))))
        print("M=%d N=%d cfg=%d (%s) res=%d  " % (M, N, cfg, names[cfg], res) + "  ".join(row), flush=True)
# floor: what a pure copy of the same bytes costs (read A [M, 320] (+ residual), write [M, 320])
x = h(M, 320); y = torch.empty_like(x)
print("copy 21 MB -> 21 MB: %.1f us" % t(lambda: y.copy_(x)), flush=True)
