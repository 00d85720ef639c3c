"""The four-phase upsampling convs (Engine.conv2d_up2x, one launch with blockIdx.z = phase) on the ping-pong tile 15 against the table's choice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402
from genima_amd.packing import pack_upsample_phases  # noqa: E402

E = Engine("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


for B, H, C, N in [(8, 256, 256, 256), (8, 128, 512, 512), (8, 64, 512, 512), (8, 32, 640, 640), (8, 16, 1280, 1280), (1, 256, 256, 256), (1, 128, 512, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").half()
    w4 = pack_upsample_phases(torch.randn(N, C, 3, 3) * (9 * C) ** -0.5).cuda()
    b = torch.zeros(N, device="cuda").half()
    fl = 2.0 * B * H * H * 4 * N * 4 * C
    line = f"up2x {B}x{H}x{H}x{C} -> {N}:"
    ms = timeit(lambda: E.conv2d_up2x(x, w4, b, name="up"))
    line += f"  table / autotune {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF/s |"
    E.no_table, E.autotune = True, False
    for rep in range(2):
        for tile in (7, 8, 9, 15):
            E.lib.gn_set_gemm_tile_override(tile - 1)
            ms = timeit(lambda: E.conv2d_up2x(x, w4, b, name="up"))
            line += f"  t{tile} {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f}"
    E.lib.gn_set_gemm_tile_override(-1)
    E.no_table, E.autotune = False, True
    print(line, flush=True)
