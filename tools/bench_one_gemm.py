"""Run ONE conv shape with ONE tile config many times (for rocprofv3 --pmc passes).  env: CFG, CIN, COUT, HW, B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genima_amd.engine import Engine  # noqa: E402

E = Engine("cuda:0")
B, cin, cout, hw, cfg = (int(os.environ.get(k, d)) for k, d in (("B", 8), ("CIN", 512), ("COUT", 512), ("HW", 128), ("CFG", 6)))
x = (torch.randn(B, hw, hw, cin, device="cuda") * 0.5).half()
w = (torch.randn(cout, 9 * cin, device="cuda") * 0.02).half()
b = torch.randn(cout, device="cuda").half()
E.lib.gn_set_gemm_tile_override(cfg)
for _ in range(10):
    E.conv2d(x, w, b)
torch.cuda.synchronize()
