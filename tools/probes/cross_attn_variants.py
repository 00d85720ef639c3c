"""The cross-attention shapes (77 prompt keys, V^T) on the block shapes of attention.hip:
variant 0 = 4 waves x 32 rows (default), 1 = 4 waves x 64 rows, 2 = 8 waves x 32 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    a, b = E.event(), E.event(); E.event_record(a)
    for _ in range(iters): fn()
    E.event_record(b); return E.event_elapsed_ms(a, b) / iters * 1e3
for B in (8, 1):
    for heads, nq in ((5, 4096), (10, 1024), (20, 256), (20, 64)):
        C = heads * 64
        q = torch.randn(B, nq, C, device="cuda").half()
        k = torch.randn(B, 77, C, device="cuda").half()
        vt = torch.randn(B, C, 128, device="cuda").half()  # V^T, key columns padded to a multiple of 64 (the block-shape variants take V^T only)
        o = torch.empty(B, nq, C, device="cuda", dtype=torch.float16)
        row = []
        for var in (0, 1, 2):
            prev = E.lib.gn_attention_set_variant(var)
            row.append(timeit(lambda: E.attention(q, k, vt, heads, Nk=77, out=o)))
            E.lib.gn_attention_set_variant(prev)
        print(f"B={B} heads={heads} Nq={nq} Nk=77: " + "  ".join(f"variant {i} {t:6.1f} us" for i, t in enumerate(row)), flush=True)
