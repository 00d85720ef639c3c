"""One GEMM shape, one tile, many launches (for rocprofv3 --pmc).  env: KIND=linear|conv, M,N,K / CIN,COUT,HW,B, CFG, RES=1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True
g = lambda k, d: int(os.environ.get(k, d))
E.lib.gn_set_gemm_tile_override(g("CFG", 9))
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
if os.environ.get("KIND", "linear") == "linear":
    M, N, K = g("M", 8192), g("N", 640), g("K", 640)
    x, w, b, r = h(M, K), h(N, K, sc=0.05), h(N), h(M, N)
    fn = lambda: E.linear(x, w, b, residual=r if g("RES", 1) else None)
else:
    B, cin, cout, hw = g("B", 8), g("CIN", 320), g("COUT", 320), g("HW", 64)
    x, w, b = h(B, hw, hw, cin), h(cout, 9 * cin, sc=0.02), h(cout)
    fn = lambda: E.conv2d(x, w, b)
for _ in range(g("ITERS", 30)): fn()
torch.cuda.synchronize()
