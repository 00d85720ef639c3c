"""Short-K Linear: time vs M (fixed overhead vs slope), tuned tile config.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.autotune = True


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


for k, n in [(320, 320), (640, 640), (1280, 1280), (320, 1280), (1280, 320)]:
    for m in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
        x = (torch.randn(m, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.5).half()
        b = torch.zeros(n, device="cuda").half()
        ms = timeit(lambda: E.linear(x, w, b))
        fl = 2.0 * m * k * n
        by = 2.0 * (m * k + m * n + n * k)
        print(f"linear M={m:6d} K={k:4d} N={n:4d}: {ms * 1000:7.1f} us  {fl / ms / 1e9:6.1f} TF/s  {by / ms / 1e6:6.0f} GB/s", flush=True)
