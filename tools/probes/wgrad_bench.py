"""gn_wgrad (natural-layout weight gradient, csrc/gemm_tn.hip) against the transposed-copy path it replaces, per hot shape of the train step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
from genima_amd import train_ops as T
E = Engine("cuda:0", autotune=True)
def h(*s): return (torch.randn(*s, device="cuda") * 0.5).half()
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = E.event(), E.event(); E.event_record(a)
    for _ in range(n): fn()
    E.event_record(b); return E.event_elapsed_ms(a, b) / n * 1e3
for (R, N, K) in ((32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (616, 1024, 1024)):
    dy, x = h(R, N), h(R, K); dw = torch.zeros(N, K, device="cuda")
    def old():
        dyt = T.transpose2d(E, dy, R, N); xt = T.transpose2d(E, x, R, K)
        T.gemm(E, dyt, xt, dw, N, K, dyt.shape[1], dyt.shape[1], dyt.shape[1], K, f32_out=True, accumulate=True)
    fl = 2.0 * R * N * K
    for tile in (1, 2, 3, 4):
        us = t(lambda: T.wgrad(E, dy, x, dw, tile=tile))
        print(f"linear R={R} N={N} K={K} tile {tile}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
    us = t(old); print(f"linear R={R} N={N} K={K} old   : {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
for (B, H, C, N) in ((8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (8, 32, 320, 640)):
    x, dy = h(B, H, H, C), h(B, H, H, N); dw = torch.zeros(N, 9 * C, device="cuda")
    R = B * H * H
    def old():
        dyt = T.transpose2d(E, dy.view(R, N), R, N); cols = T.im2col_t(E, x, 3, 1, 1)
        T.gemm(E, dyt, cols, dw, N, 9 * C, R, R, R, 9 * C, f32_out=True, accumulate=True)
    fl = 2.0 * R * N * 9 * C
    for tile in (1, 2, 3, 4):
        us = t(lambda: T.wgrad(E, dy, x, dw, ksize=3, stride=1, pad=1, tile=tile))
        print(f"conv B={B} H={H} C={C} N={N} tile {tile}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
    us = t(old); print(f"conv B={B} H={H} C={C} N={N} old   : {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
