"""Ablation timing of the ping-pong GEMM (env GN_PP_ABL=0..6, csrc/gemm_pp.hip; needs a probe build: GN_HIPCC_EXTRA=-DGN_PP_ABLATIONS python -m genima_amd.build --force): what bounds the K loop.  Results of ablated
builds are wrong by construction; only the time is read."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.no_table = True
E.lib.gn_set_gemm_tile_override(int(os.environ.get("CFG", "14")))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


def h(*shape, s=0.5):
    return (torch.randn(*shape, device="cuda") * s).half()


out = [f"ABL={os.environ.get('GN_PP_ABL', '0')} CFG={os.environ.get('CFG', '14')}"]
for cin, cout, hw in [(512, 512, 128), (320, 320, 64), (1280, 640, 32)]:
    x, w, b = h(8, hw, hw, cin), h(cout, 9 * cin, s=0.02), h(cout)
    ms = timeit(lambda: E.conv2d(x, w, b))
    out.append(f"conv{cin}->{cout}@{hw}: {2.0 * 8 * hw * hw * cout * 9 * cin / ms / 1e9:7.1f} TF")
x, w = h(16384, 4096), h(4096, 4096, s=0.05)
ms = timeit(lambda: E.linear(x, w))
out.append(f"lin16384x4096x4096: {2.0 * 16384 * 4096 * 4096 / ms / 1e9:7.1f} TF")
print(" | ".join(out), flush=True)
