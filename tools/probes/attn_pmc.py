"""One attention shape, many launches: the target of the rocprofv3 --pmc passes that decompose the kernel's wave cycles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
B, heads, n = 8, int(os.environ.get("HEADS", "5")), int(os.environ.get("N", "4096"))
C = heads * 64
qk = torch.randn(B, n, 2 * C, device="cuda").half()
vt = torch.randn(B, C, n, device="cuda").half()
o = torch.empty(B, n, C, device="cuda", dtype=torch.float16)
for _ in range(int(os.environ.get("ITERS", "20"))):
    E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, Nk=n, causal=False, out=o)
torch.cuda.synchronize()
