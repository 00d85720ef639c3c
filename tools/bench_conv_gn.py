"""Micro-benchmark of the patch-resident 3x3 conv with fused GroupNorm-apply + SiLU (csrc/conv_gn.hip) against the gn_groupnorm_fwd +
gn_gemm launches it replaces, on the VAE decoder's shapes at B = 8 tiled 512^2 (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd._lib import ACT_SILU  # noqa: E402
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0", autotune=True)
B = int(os.environ.get("B", "8"))


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    a, e = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(e)
    return E.event_elapsed_ms(a, e) / iters * 1000.0


for H, Cin, Cout in [tuple(int(v) for v in t.split("x")) for t in os.environ.get("SHAPES", "512x128x8,512x128x128,512x256x128,256x256x256,256x512x256,128x512x512,64x512x512,64x384x384").split(",")]:
    x = torch.randn(B, H, H, Cin, device="cuda").half()
    w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).half()
    bias, gamma, beta = torch.randn(Cout, device="cuda").half(), torch.ones(Cin, device="cuda").half(), torch.zeros(Cin, device="cuda").half()
    n, y0, y1 = torch.empty_like(x), torch.empty(B, H, H, Cout, device="cuda", dtype=torch.float16), torch.empty(B, H, H, Cout, device="cuda", dtype=torch.float16)
    st = E.groupnorm_stats(x, gamma, beta, 32, 1e-6)
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    t_gn = timeit(lambda: E.groupnorm(x, gamma, beta, 32, 1e-6, act=ACT_SILU, out=n))
    t_cv = timeit(lambda: E.conv2d(n, w, bias, out=y0))
    t_st = timeit(lambda: E.groupnorm_stats(x, gamma, beta, 32, 1e-6))
    t_fu = timeit(lambda: E.conv2d_gn(x, st, w, bias, out=y1))
    t_pl = timeit(lambda: E.conv2d_gn(x, None, w, bias, act=0, out=y1))
    print(f"B={B} {H}x{H} {Cin}->{Cout}: GN {t_gn:7.1f} + conv {t_cv:7.1f} ({fl / t_cv / 1e6:6.1f} TF/s) = {t_gn + t_cv:7.1f} us | stats {t_st:6.1f} + conv_gn {t_fu:7.1f} "
          f"({fl / t_fu / 1e6:6.1f} TF/s) = {t_st + t_fu:7.1f} us | patch conv without GN {t_pl:7.1f} us", flush=True)
