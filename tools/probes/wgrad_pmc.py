"""Two weight-gradient shapes, many launches: the target of rocprofv3 --pmc passes over gemm_tn_kernel (tools/probes/wgrad_pmc.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
from genima_amd import train_ops as T
E = Engine("cuda:0", autotune=True)
def h(*s): return (torch.randn(*s, device="cuda") * 0.5).half()
B, H, C, N = 8, 64, 320, 320
x, dy = h(B, H, H, C), h(B, H, H, N); dw = torch.zeros(N, 9 * C, device="cuda")
for _ in range(20):
    T.wgrad(E, dy, x, dw, ksize=3, stride=1, pad=1, tile=2)
B, H, C, N = 8, 32, 640, 640
x, dy = h(B, H, H, C), h(B, H, H, N); dw = torch.zeros(N, 9 * C, device="cuda")
for _ in range(20):
    T.wgrad(E, dy, x, dw, ksize=3, stride=1, pad=1, tile=1)
torch.cuda.synchronize()
