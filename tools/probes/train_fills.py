"""Which call sites of one ControlNet fine-tune step launch a torch fill (torch.zeros / zeros_like / Tensor.zero_ / new_zeros)?
    python tools/probes/train_fills.py            (on the GPU box; prints call site -> count for ONE steady-state step)
Used to pick what to fold into the kernels (profiles/r05_v10_train_b8_window.txt: 89 FillFunctor launches per step)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_train  # noqa: E402

sites = collections.Counter()
live = [False]


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        if live[0]:
            fr = [x for x in traceback.extract_stack()[:-1] if "genima_amd" in x.filename]
            if fr:
                sites[(name, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:][::-1]))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n)
wrap(torch.Tensor, "zero_")
wrap(torch.Tensor, "new_zeros")
wrap(torch.Tensor, "fill_")

from genima_amd.training import ControlNetTrainer  # noqa: E402

orig_step = ControlNetTrainer.train_step
count = [0]


def step(self, batch):
    count[0] += 1
    live[0] = count[0] == 3
    try:
        return orig_step(self, batch)
    finally:
        live[0] = False


ControlNetTrainer.train_step = step
args = bench_train.parse_args(["--steps", "2", "--warmup", "2"])
bench_train.run(args)
for (name, where), n in sites.most_common():
    print(f"{n:4d}  {name:10s} {where}")
print("total", sum(sites.values()))
