"""Host control flow of the fine-tune loop (genima_amd/train_loop.py) against the reference's ``main``:
lr schedules (diffusion/train_controlnet_genima.py:1206-1213 -> diffusers.optimization, pinned here by the same lambdas of the
installed ``transformers.optimization``), ``checkpoints_total_limit`` rotation (:1416-1448), ``--resume_from_checkpoint latest``
(:1281-1306), accumulation-aware global step (:1410-1414)."""
import math
import os

import pytest
import torch

from genima_amd.train_loop import TrainLoop, get_scheduler, list_checkpoints, resolve_resume, rotate_checkpoints


@pytest.mark.parametrize("name,kw", [
    ("constant", {}), ("constant_with_warmup", {}), ("linear", {}), ("cosine", {"num_cycles": 0.5}), ("cosine", {"num_cycles": 1}),
    ("cosine_with_restarts", {"num_cycles": 3}), ("polynomial", {"power": 2.0}),
])
def test_lr_schedules_match_transformers(name, kw):
    import transformers.optimization as TO

    warm, total, lr = 7, 50, 1e-5
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
    if name == "constant":
        ref = TO.get_constant_schedule(opt)
    elif name == "constant_with_warmup":
        ref = TO.get_constant_schedule_with_warmup(opt, warm)
    elif name == "linear":
        ref = TO.get_linear_schedule_with_warmup(opt, warm, total)
    elif name == "cosine":
        # diffusers' get_scheduler drops num_cycles for plain "cosine": the half cosine whatever --lr_num_cycles says
        ref = TO.get_cosine_schedule_with_warmup(opt, warm, total)
    elif name == "cosine_with_restarts":
        ref = TO.get_cosine_with_hard_restarts_schedule_with_warmup(opt, warm, total, num_cycles=kw["num_cycles"])
    else:
        ref = TO.get_polynomial_decay_schedule_with_warmup(opt, warm, total, lr_end=1e-7, power=kw["power"])
    f = get_scheduler(name, warm, total, lr_init=lr, **kw)
    for step in range(total + 5):
        want = ref.get_last_lr()[0] / lr
        assert abs(f(step) - want) < 1e-9 * max(1.0, want), (name, step, f(step), want)
        opt.step()
        ref.step()


def test_rotation_and_resume(tmp_path):
    out = str(tmp_path)
    for s in (500, 1000, 1500, 10000):
        os.makedirs(os.path.join(out, f"checkpoint-{s}", "controlnet"))
    os.makedirs(os.path.join(out, "logs"))
    assert list_checkpoints(out) == ["checkpoint-500", "checkpoint-1000", "checkpoint-1500", "checkpoint-10000"]  # numeric, not lexical
    assert resolve_resume(out, "latest") == "checkpoint-10000"
    assert resolve_resume(out, "/somewhere/else/checkpoint-1000/") == "checkpoint-1000"
    assert resolve_resume(out, None) is None and resolve_resume(str(tmp_path / "empty"), "latest") is None
    assert rotate_checkpoints(out, None) == []
    assert rotate_checkpoints(out, 5) == []                      # 4 existing + the one about to be written = 5: fits
    assert rotate_checkpoints(out, 3) == ["checkpoint-500", "checkpoint-1000"]  # keep limit - 1 = 2 before the save
    assert list_checkpoints(out) == ["checkpoint-1500", "checkpoint-10000"] and os.path.isdir(os.path.join(out, "logs"))


class _FakeTrainer:
    """ControlNetTrainer's loop-facing surface: train_step / sync_gradients / grad_accum / save_state / load_state."""

    def __init__(self, accum):
        self.grad_accum, self._micro, self.sync_gradients, self.steps, self.loaded = accum, 0, True, 0, None
        self.end_of_dataloader, self.pending = False, 0

    def train_step(self, batch):
        self._micro += 1
        self.pending += 1
        self.sync_gradients = self._micro % self.grad_accum == 0 or self.end_of_dataloader
        if self.end_of_dataloader:
            self._micro = 0
        if self.sync_gradients:
            self.pending = 0
        self.steps += int(self.sync_gradients)
        return float(batch)

    def save_state(self, out, step):
        d = os.path.join(out, f"checkpoint-{step}")
        os.makedirs(os.path.join(d, "controlnet"))
        return d

    def load_state(self, d):
        self.loaded = d
        return int(d.rsplit("-", 1)[1])


def test_loop_accumulation_checkpoints_resume(tmp_path):
    out = str(tmp_path)
    data = list(range(10))  # 10 micro-batches per epoch, 2 per optimizer step -> 5 global steps per epoch
    tr = _FakeTrainer(2)
    logs = []
    seen = []
    loop = TrainLoop(tr, out, num_train_epochs=3, checkpointing_steps=4, checkpoints_total_limit=2, validation_steps=5,
                     validate=seen.append, log=logs.append)
    assert loop.run(data) == 15 and tr.steps == 15
    assert list_checkpoints(out) == ["checkpoint-8", "checkpoint-12"]  # 4 was rotated out before 12 was written
    assert seen == [5, 10, 15]
    tr2 = _FakeTrainer(2)
    loop2 = TrainLoop(tr2, out, max_train_steps=14, checkpointing_steps=100, resume_from_checkpoint="latest", log=logs.append)
    assert loop2.run(data) == 14 and tr2.loaded.endswith("checkpoint-12") and tr2.steps == 2
    assert math.ceil(len(data) / 2) == 5 and any("Resuming from checkpoint checkpoint-12" in m for m in logs)
    tr3 = _FakeTrainer(1)
    loop3 = TrainLoop(tr3, str(tmp_path / "fresh"), max_train_steps=3, resume_from_checkpoint="latest", log=logs.append)
    assert loop3.run(data) == 3 and tr3.loaded is None and any("does not exist" in m for m in logs)


def test_cosine_default_is_the_half_cosine():
    """``--lr_scheduler cosine`` with the default ``--lr_num_cycles 1``: half a cosine to 0 at the end, not a full cycle back to peak."""
    f = get_scheduler("cosine", 0, 100)  # num_cycles left at its default (1), as the reference passes it
    assert abs(f(50) - 0.5) < 1e-12 and f(100) < 1e-12 and abs(f(25) - 0.5 * (1 + math.cos(math.pi * 0.25))) < 1e-12


def test_epoch_end_flushes_partial_accumulation(tmp_path):
    """7 micro-batches, accumulation 3: accelerate syncs at the end of the dataloader, so an epoch is ceil(7/3) = 3 optimizer steps and
    nothing is carried into the next epoch; a resumed run that is already complete trains nothing."""
    tr = _FakeTrainer(3)
    loop = TrainLoop(tr, str(tmp_path), num_train_epochs=2, checkpointing_steps=6, log=lambda m: None)
    assert loop.run(list(range(7))) == 6 and tr.steps == 6 and tr.pending == 0 and tr._micro == 0
    tr2 = _FakeTrainer(3)
    loop2 = TrainLoop(tr2, str(tmp_path), max_train_steps=6, resume_from_checkpoint="latest", log=lambda m: None)
    assert loop2.run(list(range(7))) == 6 and tr2.steps == 0 and tr2.loaded.endswith("checkpoint-6")
