"""The ping-pong tile (15) on big-M problems: time vs K at fixed (M, N) -- slope = per-K-tile cost of a round of 256 x 256 tiles, intercept = what a
round costs besides its K loop (workgroup start, first-tile fetch, epilogue drain)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def t(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = E.event(), E.event(); E.event_record(a)
        for _ in range(n): fn()
        E.event_record(b); best = min(best, E.event_elapsed_ms(a, b) / n)
    return best * 1e3
E.lib.gn_set_gemm_tile_override(14)
for (M, N) in ((131072, 512), (65536, 256), (32768, 512)):
    tiles = (M // 256) * (N // 256)
    row = []
    for K in (256, 512, 1024, 2048, 4096):
        x, w, b = h(M, K), h(N, K, sc=K ** -0.5), h(N)
        us = t(lambda: E.linear(x, w, b))
        row.append((K, us))
    (k0, u0), (k1, u1) = row[1], row[-1]
    slope = (u1 - u0) / (k1 - k0) * 64
    icpt = u0 - slope * k0 / 64
    rounds = tiles / 256
    print(f"M={M} N={N}: {tiles} tiles = {rounds:.2f} rounds | " + "  ".join(f"K={k} {u:.1f}" for k, u in row) +
          f" | {slope:.2f} us per 64-wide K tile, intercept {icpt:.1f} us = {icpt / max(rounds, 1):.1f} us per round; MFMA-only K tile at 20 ns per MFMA: {rounds * 32 * 0.0200 * 2:.2f} us", flush=True)
E.lib.gn_set_gemm_tile_override(-1)
