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
sd = _weights()
g0 = torch.Generator().manual_seed(1)
sd.update({"t.norm.weight": torch.ones(C), "t.norm.bias": torch.zeros(C), "t.proj_in.weight": (torch.randn(C, C, generator=g0) * C ** -0.5).half().float(),
           "t.proj_in.bias": torch.zeros(C)})
W = packing.pack_state_dict(sd, "cuda")
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

    Bn = max(1, M // 4096)
    x4 = a.view(Bn, M // Bn, C)
    st = E.groupnorm_stats(x4, W["t.norm.weight"], W["t.norm.bias"], 32, 1e-6)
    gn_out, pin_out = torch.empty_like(x4), torch.empty_like(x4)

    def unfused_front():
        E.groupnorm(x4, W["t.norm.weight"], W["t.norm.bias"], 32, 1e-6, out=gn_out)
        E.linear(gn_out, W["t.proj_in.weight"], W["t.proj_in.bias"], out=pin_out)
        E.linear(pin_out, W[b + ".attn1.to_qkv.ln_weight"], W[b + ".attn1.to_qkv.ln_c2"], ln_c1=W[b + ".attn1.to_qkv.ln_c1"], split_n=2 * C,
                 rows_per_batch=M // Bn, pad_cols=M // Bn)

    def fused_front():
        s2 = E.groupnorm_stats(x4, W["t.norm.weight"], W["t.norm.bias"], 32, 1e-6)
        E.tblock_front(x4, s2, W["t.tblock_front.tape"], M // Bn)

    ff = 2.0 * M * C * C * 4
    for name, fn in (("front unfused (GN + 2 launches)", unfused_front), ("front fused (stats + 1 launch)", fused_front),
                     ("front fused, chain only", lambda: E.tblock_front(x4, st, W["t.tblock_front.tape"], M // Bn))):
        us = timeit(fn)
        print(f"M={M:6d} {name:32s} {us:8.1f} us  {ff / us / 1e6:7.1f} TFLOP/s", flush=True)
    fm, ft = 2.0 * M * C * C * 2, 2.0 * M * C * C * 14
    for name, fn, fl in (("mid  unfused (2 launches)", unfused_mid, fm), ("mid  fused", lambda: E.tblock_mid(a, res, W[b + ".tblock_mid.tape"]), fm),
                         ("tail unfused (4 launches)", unfused_tail, ft), ("tail fused", lambda: E.tblock_tail(a, res, x, W[b + ".tblock_tail.tape"]), ft)):
        us = timeit(fn)
        print(f"M={M:6d} {name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
