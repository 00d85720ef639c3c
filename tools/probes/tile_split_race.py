"""(tile, K split) grid on the long-K / few-tile problems of the call: does a one-workgroup-per-CU tile (ping-pong 256x256, tile 15)
win once its K split makes the grid fit the 256 CUs?  The autotuner races K splits of the WINNING tile only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
plan = [0, 0]
orig = E._gemm
def patched(d, keep):
    d.tile, d.splitk = plan
    return orig(d, keep)
E._gemm = patched
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def timeit(fn):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(3):
        a, b = E.event(), E.event(); E.event_record(a)
        for _ in range(5): fn()
        E.event_record(b); best = min(best, E.event_elapsed_ms(a, b) / 5)
    return best
convs = [(8, 16, 1280, 1280), (8, 32, 640, 640), (8, 64, 320, 320), (8, 8, 1280, 1280), (8, 16, 640, 1280), (8, 32, 320, 640), (8, 16, 2560, 1280), (8, 32, 1280, 640)]
lins = [(8192, 5120, 640), (2048, 10240, 1280), (8192, 640, 2560), (2048, 1280, 5120), (32768, 2560, 320)]
grid = [(9, 0), (9, 2), (9, 3), (9, 4), (20, 0), (20, 2), (20, 3), (20, 4), (23, 0), (23, 2), (23, 4), (7, 0), (7, 2), (7, 3), (7, 4), (7, 6), (15, 1), (15, 2), (15, 3), (15, 4), (15, 5), (15, 6), (15, 8), (15, 12), (8, 0), (8, 2), (8, 3), (8, 4)]
for (B, hw, cin, cout) in convs:
    x, w, b = h(B, hw, hw, cin), h(cout, 9 * cin, sc=0.02), h(cout)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    res = []
    for (t, sk) in grid:
        if sk * 512 > 9 * cin: continue
        plan[:] = [t, sk]
        try: ms = timeit(lambda: E.conv2d(x, w, b))
        except Exception as ex: continue
        res.append((ms, t, sk))
    res.sort()
    print(f"conv {cin}->{cout}@{hw}x{hw} b{B}: " + "  ".join(f"t{t}/sk{sk} {ms*1e3:.1f}us {fl/ms/1e9:.0f}TF" for ms, t, sk in res[:6]), flush=True)
for (M, N, K) in lins:
    x, w, b = h(M, K), h(N, K, sc=0.05), h(N)
    fl = 2.0 * M * N * K
    res = []
    for (t, sk) in grid:
        if sk > 1 and sk * 512 > K: continue
        plan[:] = [t, sk]
        try: ms = timeit(lambda: E.linear(x, w, b))
        except Exception as ex: continue
        res.append((ms, t, sk))
    res.sort()
    print(f"linear {M}x{N}x{K}: " + "  ".join(f"t{t}/sk{sk} {ms*1e3:.1f}us {fl/ms/1e9:.0f}TF" for ms, t, sk in res[:6]), flush=True)
