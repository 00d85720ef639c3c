"""Does the GEMM rate depend on the operand VALUES (DVFS: power -> clock)?  Same kernels, zero / constant / random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.no_table = True


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


for fill in ("zeros", "const", "randn*0.02", "randn*0.5", "rand[0,1)"):
    def mk(*shape):
        if fill == "zeros":
            return torch.zeros(*shape, device="cuda", dtype=torch.float16)
        if fill == "const":
            return torch.full(shape, 0.01, device="cuda", dtype=torch.float16)
        if fill == "randn*0.02":
            return (torch.randn(*shape, device="cuda") * 0.02).half()
        if fill == "randn*0.5":
            return (torch.randn(*shape, device="cuda") * 0.5).half()
        return torch.rand(*shape, device="cuda").half()
    row = [f"{fill:12s}"]
    for cfg in (6, 14):
        E.lib.gn_set_gemm_tile_override(cfg)
        x, w, b = mk(8, 128, 128, 512), mk(512, 9 * 512), mk(512)
        ms = timeit(lambda: E.conv2d(x, w, b))
        row.append(f"cfg{cfg} conv512@128 {2.0 * 8 * 128 * 128 * 512 * 9 * 512 / ms / 1e9:7.1f}")
        x, w = mk(16384, 4096), mk(4096, 4096)
        ms = timeit(lambda: E.linear(x, w))
        row.append(f"lin16384x4096x4096 {2.0 * 16384 * 4096 * 4096 / ms / 1e9:7.1f}")
    print(" | ".join(row), flush=True)
