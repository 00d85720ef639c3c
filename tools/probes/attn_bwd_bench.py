"""Flash-attention backward micro-benchmark on the train-step shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd import train_ops as T  # noqa: E402
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
for B, heads, n in [(8, 5, 4096), (8, 10, 1024), (8, 20, 256)]:
    C = heads * 64
    q, k, v, do = (torch.randn(B, n, C, device="cuda").half() * 0.5 for _ in range(4))
    vt = v.transpose(1, 2).contiguous()
    lse = torch.empty(B, heads, n, device="cuda", dtype=torch.float32)
    o = E.attention(q, k, vt, heads, lse=lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    fn = lambda: T.attention_bwd(E, q, 0, k, 0, v, o, do, lse, heads, n, dq, dk, dv)  # noqa: E731
    for _ in range(3):
        fn()
    e0, e1 = E.event(), E.event()
    E.event_record(e0)
    for _ in range(10):
        fn()
    E.event_record(e1)
    ms = E.event_elapsed_ms(e0, e1) / 10
    print(f"attention_bwd B={B} heads={heads} N={n}: {ms * 1e3:8.1f} us  {14.0 * B * heads * n * n * 64 / ms / 1e9:7.1f} TFLOP/s (incl. transposes + delta)", flush=True)
