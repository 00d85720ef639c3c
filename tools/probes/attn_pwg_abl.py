"""Timing-only ablations of attention_pwg.hip (a library built with -DGN_PWG_ABLATIONS; results are wrong on purpose).
    touch genima_amd/csrc/attention_pwg.hip; GN_HIPCC_EXTRA=-DGN_PWG_ABLATIONS python -m genima_amd.build ; python tools/probes/attn_pwg_abl.py"""
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


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


names = {0: "whole kernel", 1: "no v_exp", 2: "no cvt / row sums", 3: "no softmax VALU", 4: "no LDS-DMA", 8: "no fragment reads", 16: "no barrier",
         28: "no DMA / reads / barrier", 31: "MFMAs only", 32: "no P.V MFMAs", 64: "no QK^T MFMAs", 96: "no MFMAs", 99: "reads / DMA / barrier only", 124: "softmax VALU only",
         2048: "row sums on the matrix pipe", 2304: "EARLY + row sums on the matrix pipe",
         768: "EARLY + one lgkmcnt(0) per pair stage", 1024: "dot2c row sums", 1280: "EARLY + dot2c", 1792: "EARLY + lgkmcnt(0) + dot2c",
         256: "EARLY reads, whole kernel", 259: "EARLY, no softmax VALU", 260: "EARLY, no LDS-DMA", 272: "EARLY, no barrier"}
for rep in range(3):
    for abl in [int(a) for a in os.environ.get('ABLS', '0,256,2048,2304').split(',')]:
        os.environ["GN_PWG_ABL"] = str(abl)
        ms = timeit(lambda: E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, out=o))
        hs = B * heads * (N // 256) / 256 * (N // 64) * 4  # half-stages per SIMD (5 rounds of 256 blocks)
        print(f"ABL {abl:3d} {names[abl]:28s} {ms * 1000:7.1f} us   {ms * 1e6 / hs:6.1f} ns per half-stage (8 MFMAs)", flush=True)
