"""Host-side control flow around ``ControlNetTrainer`` that diffusion/train_controlnet_genima.py keeps in ``main``: the learning-rate
schedule (``get_scheduler``, :1206-1213), checkpoint rotation under ``--checkpoints_total_limit`` (:1416-1457), ``--resume_from_checkpoint
latest`` (:1281-1306) and the epoch / accumulation bookkeeping (:1196-1203, :1317-1320, :1410-1414).  No device work happens here."""
from __future__ import annotations

import math
import os
import shutil
from typing import Callable, Iterable, List, Optional


# ------------------------------------------------------------------------------------------------ diffusers.optimization.get_scheduler
def get_scheduler(name: str, num_warmup_steps: int = 0, num_training_steps: Optional[int] = None, num_cycles: float = 1,
                  power: float = 1.0, lr_init: float = 1e-5, lr_end: float = 1e-7) -> Callable[[int], float]:
    """step -> learning-rate multiplier, the ``LambdaLR`` lambdas of diffusers 0.29 ``optimization.py`` (the reference passes
    ``--lr_scheduler constant`` by default; ``num_cycles`` = ``--lr_num_cycles`` (default 1), ``power`` = ``--lr_power``).
    The reference multiplies both step counts by ``accelerator.num_processes`` because accelerate advances the scheduler that many
    times per optimizer step; stepping once per optimizer step with the un-multiplied counts, as here, is the same schedule."""
    w = int(num_warmup_steps)
    T = None if num_training_steps is None else int(num_training_steps)

    def warm(step):
        return float(step) / float(max(1, w))

    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: warm(step) if step < w else 1.0
    if T is None:
        raise ValueError(f"lr scheduler {name!r} needs num_training_steps")
    if name == "linear":
        return lambda step: warm(step) if step < w else max(0.0, float(T - step) / float(max(1, T - w)))
    if name == "cosine":
        # diffusers' get_scheduler forwards num_cycles only to COSINE_WITH_RESTARTS (and power only to POLYNOMIAL): plain "cosine" runs
        # get_cosine_schedule_with_warmup with ITS default num_cycles = 0.5 -- half a cosine, down to 0 -- whatever --lr_num_cycles says
        def f(step):
            if step < w:
                return warm(step)
            prog = float(step - w) / float(max(1, T - w))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))
        return f
    if name == "cosine_with_restarts":
        def f(step):
            if step < w:
                return warm(step)
            prog = float(step - w) / float(max(1, T - w))
            if prog >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * prog) % 1.0))))
        return f
    if name == "polynomial":
        if not lr_init > lr_end:
            raise ValueError(f"lr_end ({lr_end}) must be smaller than the initial lr ({lr_init})")

        def f(step):
            if step < w:
                return warm(step)
            if step > T:
                return lr_end / lr_init
            rem = 1 - (step - w) / (T - w)
            return ((lr_init - lr_end) * rem ** power + lr_end) / lr_init
        return f
    raise ValueError(f"unknown lr scheduler {name!r}")


# ------------------------------------------------------------------------------------------------ checkpoint directories
def list_checkpoints(output_dir: str) -> List[str]:
    """``checkpoint-<step>`` directory names sorted by step (:1288-1291, :1424-1431)."""
    if not os.path.isdir(output_dir):
        return []
    dirs = [d for d in os.listdir(output_dir) if d.startswith("checkpoint")]
    return sorted(dirs, key=lambda x: int(x.split("-")[1]))


def rotate_checkpoints(output_dir: str, total_limit: Optional[int]) -> List[str]:
    """Called BEFORE a save: keep at most ``total_limit - 1`` so the new one fits (:1420-1448).  Returns what was removed."""
    if total_limit is None:
        return []
    ck = list_checkpoints(output_dir)
    removed = []
    if len(ck) >= total_limit:
        for d in ck[: len(ck) - total_limit + 1]:
            shutil.rmtree(os.path.join(output_dir, d))
            removed.append(d)
    return removed


def resolve_resume(output_dir: str, resume_from_checkpoint: Optional[str]) -> Optional[str]:
    """``--resume_from_checkpoint`` -> directory name inside ``output_dir`` or None (:1281-1300): "latest" picks the highest step; any
    other value is reduced to its basename; a missing "latest" starts a new run."""
    if not resume_from_checkpoint:
        return None
    if resume_from_checkpoint != "latest":
        return os.path.basename(resume_from_checkpoint.rstrip("/"))
    ck = list_checkpoints(output_dir)
    return ck[-1] if ck else None


class TrainLoop:
    """The reference's ``for epoch ... for step, batch ...`` loop over a ``ControlNetTrainer``: accumulation-aware global step,
    periodic ``save_state`` with rotation, resume, optional validation hook."""

    def __init__(self, trainer, output_dir: str, *, max_train_steps: Optional[int] = None, num_train_epochs: int = 1,
                 checkpointing_steps: int = 500, checkpoints_total_limit: Optional[int] = None,
                 resume_from_checkpoint: Optional[str] = None, is_main_process: bool = True,
                 validation_steps: Optional[int] = None, validate: Optional[Callable[[int], None]] = None, log=print):
        self.trainer, self.output_dir = trainer, output_dir
        self.max_train_steps, self.num_train_epochs = max_train_steps, num_train_epochs
        self.checkpointing_steps, self.total_limit = checkpointing_steps, checkpoints_total_limit
        self.resume, self.is_main = resume_from_checkpoint, is_main_process
        self.validation_steps, self.validate, self.log = validation_steps, validate, log
        self.global_step = 0

    def run(self, dataloader: Iterable, len_dataloader: Optional[int] = None) -> int:
        tr = self.trainer
        n = len_dataloader if len_dataloader is not None else len(dataloader)  # type: ignore[arg-type]
        per_epoch = math.ceil(n / tr.grad_accum)
        max_steps = self.max_train_steps if self.max_train_steps is not None else self.num_train_epochs * per_epoch
        epochs = self.num_train_epochs if self.max_train_steps is None else math.ceil(max_steps / per_epoch)
        first_epoch = 0
        path = resolve_resume(self.output_dir, self.resume)
        if self.resume and path is None:
            self.log(f"Checkpoint '{self.resume}' does not exist. Starting a new training run.")
        elif path is not None:
            self.log(f"Resuming from checkpoint {path}")
            tr.load_state(os.path.join(self.output_dir, path))
            self.global_step = int(path.split("-")[1])
            first_epoch = self.global_step // per_epoch
        if self.global_step >= max_steps:  # a resumed run that is already complete trains nothing more
            return self.global_step
        for _epoch in range(first_epoch, epochs):
            it = iter(dataloader)
            nxt = next(it, None)
            while nxt is not None:
                batch, nxt = nxt, next(it, None)
                # the last batch of an epoch syncs whatever the micro-batch count (accelerate: end_of_dataloader forces sync_gradients),
                # so an epoch makes ceil(n / accum) optimizer steps -- the ``per_epoch`` above -- and leaves nothing accumulated
                tr.end_of_dataloader = nxt is None
                try:
                    loss = tr.train_step(batch)
                finally:
                    tr.end_of_dataloader = False
                if tr.sync_gradients:
                    self.global_step += 1
                    if self.is_main and self.global_step % self.checkpointing_steps == 0:
                        rotate_checkpoints(self.output_dir, self.total_limit)
                        self.log(f"Saved state to {tr.save_state(self.output_dir, self.global_step)}")
                    if self.is_main and self.validate is not None and self.validation_steps and self.global_step % self.validation_steps == 0:
                        self.validate(self.global_step)
                self.last_loss = loss
                if self.global_step >= max_steps:
                    return self.global_step
        return self.global_step
