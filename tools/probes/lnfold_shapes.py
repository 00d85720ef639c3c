"""Folded-LayerNorm Linear against the plain Linear of the same shape and tile (what does the fold itself cost per launch?) and against
LayerNorm + plain Linear."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
from genima_amd.packing import pack_geglu
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
for (M, N, K, act, tiles) in ((32768, 960, 320, 0, (23, 20, 9)), (32768, 2560, 320, 5, (9, 7, 8)), (8192, 5120, 640, 5, (9, 7, 8)), (2048, 10240, 1280, 5, (9, 12, 16)),
                              (32768, 320, 320, 0, (23, 13)), (8192, 640, 640, 0, (10, 20, 23)), (8192, 1920, 640, 0, (7, 23, 20))):
    x, w, b = h(M, K), h(N, K, sc=K ** -0.5), h(N, sc=0.3)
    gamma, beta = (1 + 0.2 * torch.randn(K, device="cuda")).half(), h(K, sc=0.1)
    wg = (w.float() * gamma.float()[None]).half(); c1 = wg.float().sum(1).contiguous(); c2 = (w.float() @ beta.float() + b.float()).half()
    ln_us = timeit(lambda: E.layernorm(x, gamma, beta))
    out = []
    for t in tiles:
        E.lib.gn_set_gemm_tile_override(t - 1)
        p = timeit(lambda: E.linear(x, w, b, act=act))
        f = timeit(lambda: E.linear(x, wg, c2, act=act, ln_c1=c1))
        out.append(f"t{t}: plain {p:.1f} fold {f:.1f} (+{f - p:.1f})")
    print(f"{M}x{N}x{K} act {act}: LayerNorm {ln_us:.1f} us | " + " | ".join(out), flush=True)
