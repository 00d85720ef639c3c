"""Round costs of attention_pwg.hip: full rounds, one all-split round, and the mixed 2.5-round case (with / without split blocks via a child process)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


print("GN_ATTN_PWG_SPLIT =", os.environ.get("GN_ATTN_PWG_SPLIT", "(default 1)"))
for B, heads, N in [(8, 4, 4096), (8, 1, 4096), (8, 5, 4096), (8, 8, 4096), (8, 9, 4096), (1, 5, 4096), (4, 5, 4096), (8, 10, 1024), (2, 10, 1024)]:
    C = heads * 64
    qk = torch.randn(B, N, 2 * C, device="cuda").half()
    vt = torch.randn(B, C, N, device="cuda").half()
    o = torch.empty(B, N, C, device="cuda", dtype=torch.float16)
    fl = 4.0 * B * heads * N * N * 64
    line = f"B={B} heads={heads} N={N} ({B * heads * N // 256} blocks):"
    for var in (4, 5, 4, 5):
        E.lib.gn_attention_set_variant(var)
        ms = timeit(lambda: E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, out=o))
        line += f"  v{var} {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF/s"
    print(line, flush=True)
if "GN_ATTN_PWG_SPLIT" not in os.environ:
    subprocess.run([sys.executable, __file__], env=dict(os.environ, GN_ATTN_PWG_SPLIT="0"))
