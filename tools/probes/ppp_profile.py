"""Where a tile boundary of the persistent GEMM (tile 25, csrc/gemm_ppp.hip) spends its time: a library built with
GN_HIPCC_EXTRA=-DGN_PPP_PROFILE accumulates wave 0's s_memtime cycles per section and workgroup; this prints their means over the workgroups.
    GN_HIPCC_EXTRA=-DGN_PPP_PROFILE python -m genima_amd.build && python tools/probes/ppp_profile.py   (then rebuild without the define)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
names = ["K loops", "boundary -> ring requested", "plain epilogue (incl. drained wait)", "barrier behind it", "first counted wait of a K loop",
         "shared-tile epilogues", "whole kernel", "drained wait alone"]
E.lib.gn_set_gemm_tile_override(24)
for (M, N, K, res) in ((131072, 512, 1152, False), (131072, 512, 4608, False), (524288, 256, 2304, False), (40960, 1024, 640, False), (131072, 512, 1152, True)):
    x, w, b = h(M, K), h(N, K, sc=K ** -0.5), h(N)
    r = h(M, N) if res else None
    for _ in range(3): E.linear(x, w, b, residual=r)
    a, e = E.event(), E.event(); E.event_record(a)
    for _ in range(10): E.linear(x, w, b, residual=r)
    E.event_record(e); us = E.event_elapsed_ms(a, e) * 100
    buf = (C.c_uint32 * 2048)()
    assert E.lib.gn_ppp_profile_read(buf, 2048) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
    tiles = (M // 256) * (N // 256)
    clk = float(t[:, 6].mean()) / us  # cycles per us of s_memtime (100 MHz constant clock on gfx9: REFCLK) -- printed, not assumed
    print(f"M={M} N={N} K={K} residual={res}: {us:.1f} us, {tiles} tiles = {tiles / 256:.2f} per workgroup; s_memtime ticks / us = {clk:.1f}")
    for i, n in enumerate(names):
        m = float(t[:, i].mean())
        print(f"    {n:40s} {m / clk:9.2f} us per workgroup  ({m / clk / (tiles / 256):7.2f} per tile)   min {float(t[:, i].min()) / clk:8.2f} max {float(t[:, i].max()) / clk:8.2f}")
E.lib.gn_set_gemm_tile_override(-1)
