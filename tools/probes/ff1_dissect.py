"""Where the GEGLU feed-forward projections (ff.net.0: LayerNorm-folded Linear + GEGLU, C = 640 / 1280 levels) spend their time: the plain GEMM of the
same shape per tile (incl. the ping-pong tile 15, which carries neither GEGLU nor the fold), + GEGLU, + the LayerNorm fold."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
def timeit(fn):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = E.event(), E.event(); E.event_record(a)
        for _ in range(10): fn()
        E.event_record(b); best = min(best, E.event_elapsed_ms(a, b) / 10)
    return best * 1e3
for (M, N, K) in ((8192, 5120, 640), (2048, 10240, 1280), (32768, 2560, 320), (8192, 640, 3200), (2048, 1280, 6400), (8192, 1920, 640), (2048, 3840, 1280)):
    x, w, b = h(M, K), h(N, K, sc=K ** -0.5), h(N, sc=0.3)
    gamma, beta = (1 + 0.2 * torch.randn(K, device="cuda")).half(), h(K, sc=0.1)
    wg = (w.float() * gamma.float()[None]).half(); c1 = wg.float().sum(1).contiguous(); c2 = (w.float() @ beta.float() + b.float()).half()
    fl = 2.0 * M * N * K
    out = []
    for t in (7, 8, 9, 12, 14, 15):
        E.lib.gn_set_gemm_tile_override(t - 1)
        p = timeit(lambda: E.linear(x, w, b, act=0))
        s = f"t{t}: plain {p:.1f} ({fl / p / 1e6:.0f} TF/s)"
        if N >= 2560 and t != 15 and t != 14:
            g = timeit(lambda: E.linear(x, w, b, act=5))
            f = timeit(lambda: E.linear(x, wg, c2, act=5, ln_c1=c1))
            s += f" geglu {g:.1f} geglu+fold {f:.1f}"
        out.append(s)
    print(f"{M}x{N}x{K}: " + " | ".join(out), flush=True)
E.lib.gn_set_gemm_tile_override(-1)
