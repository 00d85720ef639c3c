"""Shader clock the attention_pwg.hip blocks actually run at (s_memtime / s_memrealtime inside the kernel; ablation build, GN_PWG_ABL=4096+x)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
E.lib.gn_attention_set_variant(5)
B, heads, N = 8, 10, 4096
C = heads * 64
qk = torch.randn(B, N, 2 * C, device="cuda").half()
vt = torch.randn(B, C, N, device="cuda").half()
o = torch.empty(B, N, C, device="cuda", dtype=torch.float16)
lse = torch.zeros(B, heads, N, dtype=torch.float32, device="cuda")
for abl, name in [(4096, "whole kernel"), (4096 + 8, "no fragment reads"), (4096 + 128, "reads issued, not consumed"), (4096 + 3, "no softmax VALU"), (4096, "whole kernel")]:
    os.environ["GN_PWG_ABL"] = str(abl)
    for _ in range(60):
        E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, out=o, lse=lse)
    a, bb = E.event(), E.event()
    E.event_record(a)
    for _ in range(20):
        E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, out=o, lse=lse)
    E.event_record(bb)
    us = E.event_elapsed_ms(a, bb) / 20 * 1000
    torch.cuda.synchronize()
    d = lse.view(B * heads, N // 256, 256)[:, :, 0:2].reshape(-1, 2).cpu()
    cyc, ticks = d[:, 0], d[:, 1]
    ghz = cyc / (ticks * 10.0)
    nmf = (N // 64) * 32 + 32
    print(f"{name:20s}: per block {cyc.mean():9.0f} cycles  {ticks.mean() * 0.01:6.1f} us  clock {ghz.mean():.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})  "
          f"{cyc.mean() / nmf:5.1f} cycles per MFMA   launch {us:6.1f} us", flush=True)
