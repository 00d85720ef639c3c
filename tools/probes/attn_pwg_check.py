"""attention_pwg.hip (variant 5) against attention_stream.hip (variant 4) and an fp32 torch softmax(QK^T)V on the GPU: parity, then timing.
    python tools/probes/attn_pwg_check.py            (run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
lib = E.lib


def ref(q, k, v, heads):
    B, Nq, C = q.shape
    d = C // heads
    qh, kh, vh = (t.float().view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * d ** -0.5
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Nq, C), torch.logsumexp(s, -1) * 1.4426950408889634


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = E.event(), E.event()
    E.event_record(a)
    for _ in range(iters):
        fn()
    E.event_record(b)
    return E.event_elapsed_ms(a, b) / iters


ok = True
for B, heads, N, spike in [(1, 2, 256, 0), (2, 3, 512, 0), (1, 2, 1024, 0), (2, 5, 4096, 0), (1, 3, 1024, 1), (1, 2, 320, 0), (1, 1, 128, 0), (2, 2, 1024, 2), (1, 3, 4096, 0), (3, 11, 2048, 0), (1, 2, 4096, 1)]:
    C = heads * 64
    g = torch.Generator().manual_seed(N + heads)
    qk = torch.randn(B, N, 2 * C, generator=g)
    v = torch.randn(B, N, C, generator=g)
    if spike == 1:  # one far key far above every diagonal score for head 0: forces the fallback
        d = torch.randn(64, generator=g); d = d / d.norm()
        qk[:, :, :64] += 6.0 * d
        qk[:, min(600, N - 7), C:C + 64] = 50.0 * d
    if spike == 2:  # maximum grows along the keys for head 1
        d = torch.randn(64, generator=g); d = d / d.norm()
        qk[:, :, 64:128] = 0.3 * qk[:, :, 64:128] + 8.0 * d
        qk[:, :, C + 64:C + 128] = 0.3 * qk[:, :, C + 64:C + 128] + torch.linspace(-4.0, 4.0, N)[None, :, None] * d * 3.0
    qk, v = qk.half().cuda(), v.half().cuda()
    q, k = qk[:, :, :C], qk[:, :, C:]
    vt = v.transpose(1, 2).contiguous()
    r, rl = ref(q, k, v, heads)
    outs = {}
    for var in (4, 5):
        lib.gn_attention_set_variant(var)
        lse = torch.zeros(B, heads, N, dtype=torch.float32, device="cuda")
        o = torch.full((B, N, C), float("nan"), dtype=torch.float16, device="cuda")
        E.attention(q, k, vt, heads, out=o, lse=lse)
        torch.cuda.synchronize()
        outs[var] = (o.clone(), lse.clone())
    e4, e5 = rel(outs[4][0], r), rel(outs[5][0], r)
    l5 = float((outs[5][1] - rl).abs().max())
    good = e5 < 1e-3 and l5 < 2e-2 and bool(torch.isfinite(outs[5][0]).all())
    ok &= good
    print(f"B={B} heads={heads} N={N} spike={spike}: stream {e4:.3e}  pwg {e5:.3e}  pwg-vs-stream {rel(outs[5][0], outs[4][0]):.3e}  lse err {l5:.2e}  {'ok' if good else 'FAIL'}", flush=True)
print("PARITY", "OK" if ok else "FAILED", flush=True)

for B, heads, N in [(8, 5, 4096), (8, 10, 4096), (8, 10, 1024), (4, 5, 4096), (2, 5, 4096), (1, 5, 4096), (8, 20, 1024), (1, 10, 1024)]:
    C = heads * 64
    qk = torch.randn(B, N, 2 * C, device="cuda").half()
    vt = torch.randn(B, C, N, device="cuda").half()
    o = torch.empty(B, N, C, device="cuda", dtype=torch.float16)
    fl = 4.0 * B * heads * N * N * 64
    line = f"B={B} heads={heads} N={N}:"
    for rep in range(2):
        for var in (4, 5):
            lib.gn_attention_set_variant(var)
            ms = timeit(lambda: E.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, out=o))
            line += f"  v{var} {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF/s"
    print(line, flush=True)
lib.gn_attention_set_variant(-1)
