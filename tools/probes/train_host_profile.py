"""cProfile of the host side of the ControlNet train step (where do the ~60 ms of Python per step go?)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_train
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
bench_train.run(bench_train.parse_args(["--gpus", "1", "--steps", "1", "--warmup", "2"]), quiet=True)
training.ControlNetTrainer.train_step = _step
tr, batch = holder["tr"], holder["batch"]
for _ in range(2): tr.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
