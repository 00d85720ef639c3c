"""Where is the host when the backward walk ends?  Per steady-state step of the fine-tune bench: host time of the walk, whether the GPU still
has queued work when the walk returns (host ahead) and the host time between the walk's return and the optimizer's first launch.
    python tools/probes/train_host_gap.py     (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_train  # noqa: E402
from genima_amd import train_ops as T  # noqa: E402
from genima_amd import training  # noqa: E402

rec = []
cur = {}
orig_bw = training.Graph.backward


def bw(self):
    t0 = time.perf_counter()
    orig_bw(self)
    t1 = time.perf_counter()
    cur["walk_ms"] = (t1 - t0) * 1e3
    cur["gpu_busy_at_return"] = not self.E_stream_query()
    cur["t_ret"] = t1


def q(self):
    return torch.cuda.current_stream(self.E.device).query()


training.Graph.E_stream_query = q
training.Graph.backward = bw
orig_ss = T.sumsq


def ss(E, g, out):
    t = time.perf_counter()
    if "t_ret" in cur:
        cur["ret_to_sumsq_ms"] = (t - cur.pop("t_ret")) * 1e3
    r = orig_ss(E, g, out)
    cur["gpu_busy_at_sumsq"] = not torch.cuda.current_stream(E.device).query()
    return r


T.sumsq = ss
training.T.sumsq = ss
orig_step = training.ControlNetTrainer.train_step


def step(self, batch):
    t0 = time.perf_counter()
    r = orig_step(self, batch)
    cur["host_step_ms"] = (time.perf_counter() - t0) * 1e3
    rec.append(dict(cur))
    cur.clear()
    return r


training.ControlNetTrainer.train_step = step
line = bench_train.run(bench_train.parse_args(["--steps", "6", "--warmup", "3"]))
print("ms_per_step", round(line["ms_per_step"], 2))
for r in rec[3:]:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
