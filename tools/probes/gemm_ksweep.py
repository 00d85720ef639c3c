"""Time vs K for fixed (M, N, tile): slope = per-K-tile cost, intercept = launch + prologue + epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def t(fn, n=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for (M, N) in ((8192, 640), (32768, 320), (2048, 1280)):
    for cfg in (9, 8, 12):
        E.lib.gn_set_gemm_tile_override(cfg)
        for res in (0, 1):
            row = []
            for K in (64, 128, 320, 640, 1280, 2560):
                x, w, b, r = h(M, K), h(N, K, sc=0.05), h(N), h(M, N)
                row.append("K=%d %.1f" % (K, t(lambda: E.linear(x, w, b, residual=r if res else None))))
            print("M=%d N=%d cfg=%d res=%d  " % (M, N, cfg, res) + "  ".join(row), flush=True)
