"""What the GroupNorm bridge's producer side costs a launch: the same conv / Linear with and without gn_gemm_desc.sink (eager, HIP events, 30
launches each), per shape.  usage: python tools/probes/sink_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genima_amd.engine import Engine

E = Engine("cuda:0")
E.autotune = False
g = torch.Generator().manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = E.event(), E.event()
    E.synchronize()
    E.event_record(e0)
    for _ in range(n):
        fn()
    E.event_record(e1)
    E.synchronize()
    return E.event_elapsed_ms(e0, e1) / n * 1e3


for (B, H, C, N) in [(1, 64, 320, 320), (1, 32, 640, 640), (1, 16, 1280, 1280), (1, 8, 1280, 1280), (8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280)]:
    x = (torch.randn(B, H, H, C, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, 9 * C, generator=g) * 0.02).half().cuda()
    b = torch.zeros(N).half().cuda()
    st = torch.zeros(max(1, 8 // B), B, 32, 16, dtype=torch.int64, device="cuda")  # [replicas, samples, groups, line]
    out = torch.empty(B, H, H, N, dtype=torch.float16, device="cuda")
    t0 = timeit(lambda: E.conv2d(x, w, b, out=out))
    t1 = timeit(lambda: E.conv2d(x, w, b, out=out, sink=(st, N // 32, 0, H * H)))
    print(f"conv3x3 {B}x{H}x{H}x{C}->{N}: plain {t0:7.1f} us   with sink {t1:7.1f} us   (+{t1 - t0:.1f})", flush=True)
