"""Micro-benchmark: fp8 Linear (quantise + v_mfma_scale_f32_32x32x64_f8f6f4 GEMM) vs the f16 Linear on the SDXL / SD-Turbo
transformer shapes (run on the GPU box).  Prints time and TFLOP/s of the GEMM alone and with the activation quantisation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.autotune = True


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


for m, k, n in [(32768, 640, 640), (32768, 640, 5120), (32768, 2560, 640), (8192, 1280, 1280), (8192, 1280, 10240), (8192, 5120, 1280),
                (8192, 2048, 1280), (32768, 1280, 1280), (65536, 4096, 4096)]:
    x = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    b = torch.zeros(n, device="cuda").half()
    wq, ws = E.quantize_fp8(w)
    xq, xs = E.quantize_fp8(x)
    t16 = timeit(lambda: E.linear(x, w, b))
    t8 = timeit(lambda: E.linear_fp8(xq, xs, wq, ws, b))
    tq = timeit(lambda: E.quantize_fp8(x))
    fl = 2.0 * m * k * n
    print(f"M={m:6d} K={k:5d} N={n:5d}: f16 {t16 * 1e3:7.1f} us {fl / t16 / 1e9:6.0f} TF | fp8 gemm {t8 * 1e3:7.1f} us {fl / t8 / 1e9:6.0f} TF | "
          f"quantise {tq * 1e3:6.1f} us ({2.0 * m * k * 1.5 / tq / 1e6:5.0f} GB/s) | fp8 total speedup {t16 / (t8 + tq):4.2f}x", flush=True)
