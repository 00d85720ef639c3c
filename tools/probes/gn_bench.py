"""GroupNorm(+SiLU) micro-benchmark on the hot-path shapes (run on the GPU box; GN_GROUPNORM_FUSED=<KB> moves the fused-path limit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
for b, hw, c in [(8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280), (8, 4096, 640), (8, 1024, 1280), (8, 1024, 1920), (8, 256, 2560), (1, 1024, 320), (1, 4096, 320), (8, 16384, 512), (8, 262144, 128)]:
    side = int(hw ** 0.5)
    x = torch.randn(b, side, side, c, device="cuda").half()
    g, bt = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
    for _ in range(3):
        E.groupnorm(x, g, bt, 32, 1e-5, act=1)
    e0, e1 = E.event(), E.event()
    E.event_record(e0)
    for _ in range(20):
        E.groupnorm(x, g, bt, 32, 1e-5, act=1)
    E.event_record(e1)
    ms = E.event_elapsed_ms(e0, e1) / 20
    print(f"groupnorm {b}x{hw}x{c}: {ms * 1e3:7.1f} us  {4.0 * b * hw * c / ms / 1e6:6.0f} GB/s", flush=True)
