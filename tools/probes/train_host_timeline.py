"""Where the host is relative to the GPU at the seams of the ControlNet fine-tune step (run on the GPU box):
    python tools/probes/train_host_timeline.py [--steps 6]
Prints, per timed step, host milliseconds spent launching the front + forward, the backward and the optimizer, and whether the GPU had
already drained the engine's stream when the host reached the optimizer (then the end of the backward is host-bound)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_train  # noqa: E402
from genima_amd import training  # noqa: E402

rows, cur = [], {}
_fb, _opt, _bw = training.ControlNetTrainer.forward_backward, training.ControlNetTrainer.optimizer_step, training.Graph.backward


def fb(self, *a, **k):
    cur["fb0"] = time.perf_counter()
    r = _fb(self, *a, **k)
    cur["fb1"] = time.perf_counter()
    return r


def bw(self):
    cur["bw0"] = time.perf_counter()
    cur["drained_at_bw"] = bool(self.E.stream.query())
    return _bw(self)


def opt(self):
    cur["opt0"] = time.perf_counter()
    cur["drained_at_opt"] = bool(self.E.stream.query())
    r = _opt(self)
    cur["opt1"] = time.perf_counter()
    rows.append(dict(cur))
    return r


training.ControlNetTrainer.forward_backward, training.ControlNetTrainer.optimizer_step, training.Graph.backward = fb, opt, bw
args = bench_train.parse_args(sys.argv[1:])
line = bench_train.run(args)
print("ms_per_step", line["ms_per_step"])
prev = None
for r in rows[-args.steps:]:
    print("host ms: since previous optimizer end %6.2f | forward launch %6.2f | backward launch %6.2f | optimizer launch %5.2f | drained at backward %s, at optimizer %s"
          % ((r["fb0"] - prev) * 1e3 if prev else -1.0, (r["bw0"] - r["fb0"]) * 1e3, (r["opt0"] - r["bw0"]) * 1e3, (r["opt1"] - r["opt0"]) * 1e3,
             r["drained_at_bw"], r["drained_at_opt"]))
    prev = r["opt1"]
