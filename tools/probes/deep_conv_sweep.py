"""The deepest-level 3x3 convs (8 x 8 latent, 1280 -> 1280: 29.5 MB of weights under 64 (B = 1) or 512 (B = 8) rows): main launch + split-K reduce for
every ring / DMA tile and K split -- what does the weight stream reach?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def timeit(fn, n=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = E.event(), E.event(); E.event_record(a)
        for _ in range(n): fn()
        E.event_record(b); best = min(best, E.event_elapsed_ms(a, b) / n)
    return best * 1e3
for B, cin in ((1, 1280), (8, 1280), (1, 2560), (8, 2560)):
    x, w, b = h(B, 8, 8, cin), h(1280, 9 * cin, sc=0.02), h(1280)
    wb = 1280 * 9 * cin * 2
    res = []
    for tile in (3, 10, 11, 17, 18, 9, 16, 23, 20, 21):
        for sk in (4, 8, 12, 16, 24, 32, 48):
            if sk * 256 > 9 * cin: continue
            E.lib.gn_set_gemm_tile_override(tile - 1)
            try:
                us = timeit(lambda: E.conv2d(x, w, b, splitk=sk))
            except Exception as e:  # noqa: BLE001
                continue
            res.append((us, tile, sk))
    E.lib.gn_set_gemm_tile_override(-1)
    E.no_table, E.autotune = False, True
    tab = timeit(lambda: E.conv2d(x, w, b))
    E.no_table, E.autotune = True, False
    res.sort()
    print(f"conv3x3 B={B} 8x8 {cin}->1280 (weights {wb / 1e6:.1f} MB): table {tab:.1f} us = {wb / tab / 1e6:.2f} TB/s | best " +
          " | ".join(f"t{t} sk{s} {u:.1f} us ({wb / u / 1e6:.2f} TB/s)" for u, t, s in res[:6]), flush=True)
