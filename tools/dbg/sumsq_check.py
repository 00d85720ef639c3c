import torch, sys
sys.path.insert(0, "/root/repo")
from genima_amd.engine import Engine
from genima_amd import train_ops as T
E = Engine("cuda:0")
for n in (5000, 262144, 7_500_000, 7_500_003):
    x = torch.randn(n, device="cuda") * 100
    ss = torch.zeros(1, device="cuda")
    T.sumsq(E, x, ss)
    ref = float((x.double() ** 2).sum())
    print(n, float(ss), ref, abs(float(ss) - ref) / ref)
