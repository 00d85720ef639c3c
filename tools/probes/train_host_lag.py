"""Is the host behind the GPU at the end of the train step's backward?  When ControlNetTrainer.optimizer_step is entered, ask every stream of the
step whether its queue is already empty (stream.query()): True = the GPU has caught up with the host and idles until sumsq is issued.  Also the host
time of the forward + backward walk's issue against the step time."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench_train
from genima_amd.training import ControlNetTrainer

log = []
orig_opt, orig_fb = ControlNetTrainer.optimizer_step, ControlNetTrainer.forward_backward


def fb(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_fb(self, *a, **k)
    self._fb_host = time.perf_counter() - t0
    return r


def opt(self):
    streams = [self.E.stream] + [s for s in (getattr(self, n, None) for n in ("_fwd_stream", "_wgrad_stream", "_front_stream")) if s is not None]
    idle = [bool(s.query()) for s in streams]
    t0 = time.perf_counter()
    r = orig_opt(self)
    log.append((idle, getattr(self, "_fb_host", 0.0) * 1e3, (time.perf_counter() - t0) * 1e3))
    return r


ControlNetTrainer.forward_backward, ControlNetTrainer.optimizer_step = fb, opt
sys.argv = [sys.argv[0], "--steps", "10", "--warmup", "4"]
line = bench_train.run(bench_train.parse_args())
print("ms_per_step", round(line["ms_per_step"], 2))
for idle, fbh, oh in log[4:]:
    print("streams already empty at optimizer_step [main, fwd, wgrad, front]:", idle, f"| host issue of forward+backward {fbh:.1f} ms, of the optimizer {oh:.2f} ms")
