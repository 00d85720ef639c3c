"""LayerNorm micro-benchmark on the hot-path shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
for m, c in [(32768, 320), (8192, 640), (2048, 1280), (616, 1024), (512, 1280), (32768, 640), (8192, 1280), (8192, 2048)]:
    x = torch.randn(m, c, device="cuda").half()
    g, b = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
    for _ in range(5):
        E.layernorm(x, g, b, 1e-5)
    e0, e1 = E.event(), E.event()
    E.event_record(e0)
    for _ in range(50):
        E.layernorm(x, g, b, 1e-5)
    E.event_record(e1)
    ms = E.event_elapsed_ms(e0, e1) / 50
    print(f"layernorm {m}x{c}: {ms * 1e3:6.1f} us  {4.0 * m * c / ms / 1e6:6.0f} GB/s", flush=True)
