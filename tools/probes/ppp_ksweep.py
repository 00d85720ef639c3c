"""Tile 25 (persistent skewed ping-pong, csrc/gemm_ppp.hip) against tile 15 (one launch round per 256 tiles, csrc/gemm_pp.hip): time vs K at fixed
(M, N) -- slope = per-K-tile cost of a tile, intercept / rounds = what a tile costs BESIDE its K loop (<= 6 us per tile asked).
GN_PPP_SKEW=0 in the environment: the same kernel with every workgroup starting at K = 0 (the chip-wide bursts back)."""
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
shapes = ((131072, 512), (65536, 256), (40960, 1024), (20480, 1024), (524288, 256))
for tile in (15, 25):
    E.lib.gn_set_gemm_tile_override(tile - 1)
    print(f"tile {tile}  GN_PPP_SKEW={os.environ.get('GN_PPP_SKEW', '1')}")
    for (M, N) in shapes:
        tiles = (M // 256) * (N // 256)
        row = []
        for K in (256, 512, 1024, 2048, 4096):
            if M * K * 2 > 3 << 30: continue
            x, w, b = h(M, K), h(N, K, sc=K ** -0.5), h(N)
            us = t(lambda: E.linear(x, w, b))
            row.append((K, us))
        (k0, u0), (k1, u1) = row[1], row[-1]
        slope = (u1 - u0) / (k1 - k0) * 64
        icpt = u0 - slope * k0 / 64
        rounds = tiles / 256
        tf = 2.0 * M * N * row[-1][0] / row[-1][1] / 1e6
        print(f"  M={M} N={N}: {tiles} tiles = {rounds:.2f} rounds | " + "  ".join(f"K={k} {u:.1f}" for k, u in row) +
              f" | {slope:.2f} us per 64-wide K tile, intercept {icpt:.1f} us = {icpt / max(rounds, 1):.1f} us per round; {tf:.0f} TF/s at K={row[-1][0]}", flush=True)
# the feed-forward projections of the B = 8 call (LayerNorm fold + GEGLU): the table's tile, tile 9 and tile 25
from genima_amd.packing import pack_geglu
from genima_amd.engine import _tune_table
for (M, Nh, K) in ((8192, 2560, 640), (2048, 5120, 1280)):
    x = h(M, K); w = h(2 * Nh, K, sc=K ** -0.5); b = h(2 * Nh); c1 = w.float().sum(1).contiguous()
    row = []
    for tile in (0, 9, 25):
        if tile: E.lib.gn_set_gemm_tile_override(tile - 1); E.no_table = True
        else: E.lib.gn_set_gemm_tile_override(-1); E.no_table = False
        row.append((tile, t(lambda: E.linear(x, w, b, ln_c1=c1, act=5))))
    print(f"FF {M}x{2 * Nh}x{K}: " + "  ".join(f"tile {tl if tl else 'table'} {u:.1f} us ({2.0 * M * 2 * Nh * K / u / 1e6:.0f} TF/s)" for tl, u in row), flush=True)
E.lib.gn_set_gemm_tile_override(-1)
print("timeouts", int(E.lib.gn_ppp_timeouts()))
