"""How long does the HOST take to enqueue one ControlNet train step (no synchronisation inside the loop) against the GPU time per step?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_train
args = bench_train.parse_args(["--gpus", "1", "--steps", "1", "--warmup", "1"])
import types
# reuse bench_train's setup by running it once for warm-up, then grab the trainer through a hook
orig = bench_train.ControlNetTrainer if hasattr(bench_train, "ControlNetTrainer") else None
from genima_amd import training
holder = {}
_init = training.ControlNetTrainer.__init__
def init(self, *a, **k):
    _init(self, *a, **k); holder["tr"] = self
training.ControlNetTrainer.__init__ = init
_step = training.ControlNetTrainer.train_step
def step(self, batch):
    holder["batch"] = batch
    return _step(self, batch)
training.ControlNetTrainer.train_step = step
bench_train.run(args, quiet=True)
tr, batch = holder["tr"], holder["batch"]
for _ in range(3): tr.train_step(batch)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter(); host = []
for _ in range(N):
    a = time.perf_counter(); tr.train_step(batch); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue per step: median {sorted(host)[N // 2] * 1e3:.1f} ms (min {min(host) * 1e3:.1f}, max {max(host) * 1e3:.1f}); wall per step incl. final drain {(t2 - t0) / N * 1e3:.1f} ms; drain {1e3 * (t2 - t1):.1f} ms")
