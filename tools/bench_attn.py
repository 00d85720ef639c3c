"""Micro-benchmark of the flash-attention kernel on the hot-path shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
B = int(os.environ.get("B", "8"))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


for heads, nq, nk, causal in [(5, 4096, 4096, False), (10, 1024, 1024, False), (20, 256, 256, False), (20, 64, 64, False),
                              (5, 4096, 77, False), (16, 77, 77, True)]:
    C = heads * 64
    qk = (torch.randn(B, nq, 2 * C, device="cuda")).half()
    k = qk[:, :, C:] if nk == nq else torch.randn(B, nk, C, device="cuda").half()
    vt = torch.randn(B, C, (nk + 63) // 64 * 64, device="cuda").half()
    o = torch.empty(B, nq, C, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: E.attention(qk[:, :, :C], k, vt, heads, Nk=nk, causal=causal, out=o))
    fl = 4.0 * B * heads * nq * nk * 64 * (0.5 if causal else 1.0)
    print(f"attention B={B} heads={heads} Nq={nq} Nk={nk} causal={causal}: {ms * 1000:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # the same problem with V row-major (a column slice of the q | k | v projection's output; transposed out of LDS in the kernel)
    qkv = torch.randn(B, nk, 3 * C, device="cuda").half()
    v = qkv[:, :, 2 * C:]
    o2 = torch.empty_like(o)
    ms2 = timeit(lambda: E.attention(qk[:, :, :C], k, v, heads, Nk=nk, causal=causal, out=o2, v_rowmajor=True))
    E.attention(qk[:, :, :C], k, v.transpose(1, 2).contiguous() if nk % 64 == 0 else torch.nn.functional.pad(v.transpose(1, 2), (0, (nk + 63) // 64 * 64 - nk)).contiguous(),
                heads, Nk=nk, causal=causal, out=o)
    same = torch.equal(o, o2)
    print(f"          row-major V:                                    {ms2 * 1000:8.1f} us  {fl / ms2 / 1e9:7.1f} TFLOP/s  bitwise {'==' if same else '!='} V^T path", flush=True)
