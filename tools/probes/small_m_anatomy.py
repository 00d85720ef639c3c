"""Where a small-M Linear's time goes: ONE launch between two HIP events (not a pipelined loop), median of 30, against K and against the
tile, with a trivial elementwise launch as the floor.  M = 64 / 256 rows, N = 1280 (the 8 x 8 / 16 x 16 latent levels of a B = 1 call)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.autotune = False; E.no_table = True
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def once(fn, reps=30):
    ts = []
    for _ in range(reps):
        a, b = E.event(), E.event()
        E.synchronize(); E.event_record(a); fn(); E.event_record(b); E.synchronize()
        ts.append(E.event_elapsed_ms(a, b) * 1e3); E.lib.gn_event_destroy(a); E.lib.gn_event_destroy(b)
    return statistics.median(ts)
x0 = h(64, 1280); y0 = torch.empty_like(x0)
print(f"floor: add of 64 x 1280: {once(lambda: E.add(x0, x0, out=y0)):.1f} us; empty event pair: {once(lambda: None):.1f} us")
for M in (64, 256):
    for N in (1280,):
        print(f"M={M} N={N}: K ->", end="")
        for K in (64, 256, 640, 1280, 2560, 5120):
            x, w, b = h(M, K), h(N, K, sc=0.03), h(N)
            res = []
            for tile in (18, 11, 17):
                E.lib.gn_set_gemm_tile_override(tile - 1)
                out = torch.empty(M, N, device="cuda", dtype=torch.float16)
                res.append(once(lambda: E.linear(x, w, b, out=out, splitk=1)))
            E.lib.gn_set_gemm_tile_override(-1)
            print(f"  {K}: " + "/".join(f"{r:.1f}" for r in res), end="")
        print("   (us; tiles 18 / 11 / 17)")
