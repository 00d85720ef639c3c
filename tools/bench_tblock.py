"""Micro-benchmark of the fused transformer-block chains (csrc/tblock.hip) against the gn_gemm launches they replace (run on the GPU box).
    python tools/bench_tblock.py            # M = 32768 (B = 8 tiled 512^2 at the 64x64-latent level) and M = 4096 (B = 1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from genima_amd import packing  # noqa: E402
from genima_amd._lib import ACT_GEGLU  # noqa: E402
from genima_amd.engine import Engine  # noqa: E402
from test_tblock_gpu import C, _weights  # noqa: E402

E = Engine("cuda:0", autotune=True)
W = packing.pack_state_dict(_weights(), "cuda")
b = "t.transformer_blocks.0"


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    a, e = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(e)
    return E.event_elapsed_ms(a, e) / iters * 1000.0


for M in [int(m) for m in os.environ.get("MS", "32768,4096").split(",")]:
    a, res, x = (torch.randn(M, C, device="cuda").half() for _ in range(3))
    h1, q, out = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    hid = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    h2, h3 = torch.empty_like(a), torch.empty_like(a)

    def unfused_mid():
        E.linear(a, W[b + ".attn1.to_out.0.weight"], W[b + ".attn1.to_out.0.bias"], residual=res, out=h1)
        E.linear(h1, W[b + ".attn2.to_q.ln_weight"], W[b + ".attn2.to_q.ln_c2"], ln_c1=W[b + ".attn2.to_q.ln_c1"], out=q)

    def unfused_tail():
        E.linear(a, W[b + ".attn2.to_out.0.weight"], W[b + ".attn2.to_out.0.bias"], residual=res, out=h2)
        E.linear(h2, W[b + ".ff.net.0.proj.ln_weight"], W[b + ".ff.net.0.proj.ln_c2"], ln_c1=W[b + ".ff.net.0.proj.ln_c1"], act=ACT_GEGLU, out=hid)
        E.linear(hid, W[b + ".ff.net.2.weight"], W[b + ".ff.net.2.bias"], residual=h2, out=h3)
        E.linear(h3, W["t.proj_out.weight"], W["t.proj_out.bias"], residual=x, out=out)

    fm, ft = 2.0 * M * C * C * 2, 2.0 * M * C * C * 14
    for name, fn, fl in (("mid  unfused (2 launches)", unfused_mid, fm), ("mid  fused", lambda: E.tblock_mid(a, res, W[b + ".tblock_mid.tape"]), fm),
                         ("tail unfused (4 launches)", unfused_tail, ft), ("tail fused", lambda: E.tblock_tail(a, res, x, W[b + ".tblock_tail.tape"]), ft)):
        us = timeit(fn)
        print(f"M={M:6d} {name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
