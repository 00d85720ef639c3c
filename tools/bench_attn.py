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
                              (5, 4096, 77, False), (10, 1024, 77, False), (20, 256, 77, False), (16, 77, 77, True)]:
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

# ---- fp8 (e4m3) attention: operands made by gn_attention_fp8_quantize, both products on the K = 64 fp8 MFMA -------------------------
import ctypes as C  # noqa: E402
from genima_amd._lib import AttnDesc, check  # noqa: E402

for heads, n in [(5, 4096), (10, 4096), (10, 1024), (20, 1024), (20, 256)]:
    Cc = heads * 64
    qkv = torch.randn(B, n, 3 * Cc, device="cuda").half()
    q, k, v = qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:]
    o = torch.empty(B, n, Cc, device="cuda", dtype=torch.float16)
    q8 = torch.empty(B, n, Cc, dtype=torch.uint8, device="cuda"); k8 = torch.empty_like(q8)
    v8t = torch.empty(B, Cc, n, dtype=torch.uint8, device="cuda")
    quant = lambda: check(E.lib.gn_attention_fp8_quantize(E._ctx, q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(1), k.stride(1), v.stride(1),
                                                          q.stride(0), k.stride(0), v.stride(0), B, n, heads, 0.125, q8.data_ptr(), k8.data_ptr(), v8t.data_ptr(), n), "quantize")
    d = AttnDesc()
    d.q, d.k, d.vt, d.o = q8.data_ptr(), k8.data_ptr(), v8t.data_ptr(), o.data_ptr()
    d.q_bs, d.k_bs, d.vt_bs, d.o_bs = q8.stride(0), k8.stride(0), v8t.stride(0), o.stride(0)
    d.q_rs, d.k_rs, d.vt_rs, d.o_rs = q8.stride(1), k8.stride(1), v8t.stride(1), o.stride(1)
    d.B, d.heads, d.Nq, d.Nk, d.D, d.causal, d.scale = B, heads, n, n, 64, 0, 1.0
    quant()
    t_q = timeit(quant)
    t_a = timeit(lambda: check(E.lib.gn_attention_fp8_fwd(E._ctx, C.byref(d)), "fp8 fwd"))
    t_16 = timeit(lambda: E.attention(q, k, v, heads, out=o, v_rowmajor=True))
    fl = 4.0 * B * heads * n * n * 64
    print(f"fp8 attention B={B} heads={heads} N={n}: kernel {t_a * 1000:8.1f} us {fl / t_a / 1e9:7.1f} TFLOP/s | operands {t_q * 1000:6.1f} us | "
          f"kernel + operands {fl / (t_a + t_q) / 1e9:7.1f} TFLOP/s | f16 kernel (row-major V) {t_16 * 1000:8.1f} us {fl / t_16 / 1e9:7.1f} TFLOP/s", flush=True)
