"""Training data path of the ControlNet fine-tune: RLBench PNG reader, the trainer's
pre-processing and collate, and the hand-over to the device.

Reference: ``diffusion/rlbench_dataset/rlbench_dataset.py:70-210`` (a HF ``datasets`` builder yielding {text, image, conditioning_image}),
``diffusion/train_controlnet_genima.py:870-964`` (``tokenize_captions``, Resize(bilinear) -> CenterCrop -> ToTensor (-> Normalize(0.5,
0.5) for the target), ``collate_fn`` stacking float32 NCHW tensors) and ``:775-830`` (``augment_data``, on the device here:
genima_amd/augment.py).

MI355X-first hand-over: the host decodes, resizes and crops to **uint8 NHWC** and nothing more; ``to_device`` uploads the bytes (a
quarter of the float32 NCHW volume over PCIe) and ``ToTensor`` / ``Normalize`` / NCHW -> 8-channel NHWC f16 happen in ONE HIP kernel
per tensor (``gn_image_u8_to_f16``) -- the layout ``ControlNetTrainer.train_step`` consumes directly.  ``collate_fn`` keeps the
reference's float NCHW contract for callers that want it (numpy; no torch compute).

Reference quirks reproduced on purpose: the tiled caption is the truncated string
``"tiled perspectives of a robot "`` (the task description sits in a dangling f-string statement that is evaluated -- it advances
numpy's global RNG through ``np.random.choice`` -- and discarded), and the tiled reader drops the last frame of every episode.
"""
from __future__ import annotations

import io
import os
import pickle
import random
import re
import threading
from queue import Queue
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch


def _natural_key(s: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


class RLBenchDataset:
    """Index of (image, conditioning image, caption) examples in the on-disk layout ``render_data.py`` writes:
    ``<data_path>/<task>/variation<k>/episodes/<episode>/{rgb, rgb_rendered}/<i>.png`` (tiled) or ``<camera>_{rgb, rgb_rendered}`` (per
    camera), ``variation_descriptions.pkl`` beside ``episodes``.  Same constructor knobs as the reference's ``RLBenchConfig``."""

    def __init__(self, data_path: str = "/tmp/rlbench_dataset/", tasks: str = "take_lid_off_saucepan", variation: int = 0,
                 num_demos: int = 50, cameras: str = "wrist,front,left_shoulder,right_shoulder", image_type: str = "rgb_rendered",
                 conditioning_image_type: str = "rgb", tiled: bool = True, predict_future: bool = False, predict_future_horizon: int = 20):
        self.examples: List[Dict[str, str]] = []
        for task in tasks.split(","):
            var = f"variation{variation}"
            eps_folder = os.path.join(data_path, task, var, "episodes")
            with open(os.path.join(data_path, task, var, "variation_descriptions.pkl"), "rb") as f:
                descriptions = pickle.load(f)
            demos = [d for d in os.listdir(eps_folder) if os.path.isdir(os.path.join(eps_folder, d))]
            selected = sorted(os.listdir(eps_folder), key=_natural_key)[: min(len(demos), num_demos)]
            for ep in selected:
                views = [None] if tiled else cameras.split(",")
                for cam in views:
                    pre = "" if cam is None else cam + "_"
                    rgb_path = os.path.join(eps_folder, ep, pre + conditioning_image_type)
                    render_path = os.path.join(eps_folder, ep, pre + image_type)
                    np.random.choice(descriptions)  # evaluated and discarded by the reference too (rlbench_dataset.py:118-119, :170-171)
                    text = "tiled perspectives of a robot " if tiled else "a robot arm executing '"
                    n = len([f for f in os.listdir(render_path) if ".png" in f]) - (1 if tiled else 0)  # tiled: last frame dropped (:121-123)
                    for i in range(n):
                        j = min(i + predict_future_horizon, n - 1) if predict_future else i
                        self.examples.append({"text": text, "image": os.path.join(render_path, f"{j}.png"),
                                              "conditioning_image": os.path.join(rgb_path, f"{i}.png")})

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i: int) -> Dict[str, object]:
        ex = self.examples[i]
        out = {"text": ex["text"]}
        for k in ("image", "conditioning_image"):
            with open(ex[k], "rb") as f:
                out[k] = {"path": ex[k], "bytes": f.read()}
        return out


def resize_center_crop_u8(png_bytes_or_image, resolution: int) -> np.ndarray:
    """``image.convert("RGB")`` -> Resize(resolution, BILINEAR) (shorter side, aspect kept, torchvision's size rule) -> CenterCrop ->
    uint8 HWC (train_controlnet_genima.py:895-915 up to, not including, ToTensor)."""
    from PIL import Image

    im = png_bytes_or_image
    if isinstance(im, dict):
        im = im["bytes"]
    if isinstance(im, (bytes, bytearray)):
        im = Image.open(io.BytesIO(im))
    im = im.convert("RGB")
    w, h = im.size
    if min(w, h) != resolution:
        if w <= h:
            nw, nh = resolution, int(resolution * h / w)
        else:
            nw, nh = int(resolution * w / h), resolution
        im = im.resize((nw, nh), Image.BILINEAR)
        w, h = im.size
    left, top = int(round((w - resolution) / 2.0)), int(round((h - resolution) / 2.0))
    return np.asarray(im.crop((left, top, left + resolution, top + resolution)), dtype=np.uint8)


def tokenize_captions(captions: Sequence, tokenizer, proportion_empty_prompts: float = 0.0, is_train: bool = True) -> torch.Tensor:
    """train_controlnet_genima.py:870-891 (python's ``random`` for the empty-prompt draw and the multi-caption choice)."""
    out = []
    for c in captions:
        if random.random() < proportion_empty_prompts:
            out.append("")
        elif isinstance(c, str):
            out.append(c)
        elif isinstance(c, (list, np.ndarray)):
            out.append(random.choice(c) if is_train else c[0])
        else:
            raise ValueError("caption column should contain either strings or lists of strings")
    return tokenizer(out, max_length=tokenizer.model_max_length, padding="max_length", truncation=True, return_tensors="pt").input_ids


def collate_u8(examples: Sequence[Dict], tokenizer, resolution: int, proportion_empty_prompts: float = 0.0) -> Dict[str, torch.Tensor]:
    """Host half of preprocess_train + collate: uint8 NHWC stacks + token ids (pinned when a GPU is present)."""
    px = np.stack([resize_center_crop_u8(e["image"], resolution) for e in examples])
    cd = np.stack([resize_center_crop_u8(e["conditioning_image"], resolution) for e in examples])
    ids = tokenize_captions([e["text"] for e in examples], tokenizer, proportion_empty_prompts)
    batch = {"pixel_values_u8": torch.from_numpy(px), "conditioning_pixel_values_u8": torch.from_numpy(cd), "input_ids": ids}
    if torch.cuda.is_available():
        batch = {k: v.pin_memory() for k, v in batch.items()}
    return batch


def collate_fn(examples: Sequence[Dict], tokenizer, resolution: int, proportion_empty_prompts: float = 0.0) -> Dict[str, torch.Tensor]:
    """The reference's batch contract (train_controlnet_genima.py:934-964): float32 NCHW ``pixel_values`` in [-1, 1] (ToTensor +
    Normalize([0.5], [0.5])), ``conditioning_pixel_values`` in [0, 1], int64 ``input_ids`` [b, 77]."""
    b = collate_u8(examples, tokenizer, resolution, proportion_empty_prompts)
    px = b["pixel_values_u8"].numpy().astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
    cd = b["conditioning_pixel_values_u8"].numpy().astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
    return {"pixel_values": torch.from_numpy(np.ascontiguousarray((px - np.float32(0.5)) / np.float32(0.5))),
            "conditioning_pixel_values": torch.from_numpy(np.ascontiguousarray(cd)), "input_ids": b["input_ids"]}


def to_device(E, batch_u8: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """uint8 NHWC host batch -> what ``ControlNetTrainer.train_step`` takes: f16 NHWC 8-channel tensors on the device, ToTensor +
    Normalize fused into the byte -> half conversion kernel (target: v / 255 * 2 - 1, conditioning: v / 255)."""
    dev = E.device
    px = batch_u8["pixel_values_u8"].to(dev, non_blocking=True)
    cd = batch_u8["conditioning_pixel_values_u8"].to(dev, non_blocking=True)
    return {"pixel_values": E.image_u8_to_f16(px, 8, 2.0, -1.0), "conditioning_pixel_values": E.image_u8_to_f16(cd, 8, 1.0, 0.0),
            "input_ids": batch_u8["input_ids"].to(dev, non_blocking=True)}


class DataLoader:
    """``torch.utils.data.DataLoader(train_dataset, shuffle=True, collate_fn=collate_fn, batch_size=, num_workers=)``
    (train_controlnet_genima.py:1187-1193) for the uint8 path: seeded shuffle per epoch, last partial batch kept, and a background
    thread that decodes / resizes ``prefetch`` batches ahead so PNG decoding overlaps the device step (the reference's default
    ``num_workers=0`` decodes inline and can starve 8 GPUs)."""

    def __init__(self, dataset, batch_size: int, tokenizer, resolution: int, shuffle: bool = True, seed: int = 0, prefetch: int = 2,
                 proportion_empty_prompts: float = 0.0, rank: int = 0, world: int = 1):
        self.ds, self.bs, self.tok, self.res = dataset, batch_size, tokenizer, resolution
        self.shuffle, self.seed, self.prefetch, self.pep = shuffle, seed, max(0, prefetch), proportion_empty_prompts
        self.rank, self.world, self.epoch = rank, world, 0

    def __len__(self):
        n_batches = (len(self.ds) + self.bs - 1) // self.bs
        return (n_batches + self.world - 1) // self.world  # the same on every rank (a shorter rank would leave the others in a collective)

    def _batches(self) -> List[List[int]]:
        idx = list(range(len(self.ds)))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(idx)
        if self.world > 1 and idx:
            # data parallel, as accelerate's prepared loader shards (BatchSamplerShard, even_batches=True): the epoch's permutation is
            # cut into batches, batch k goes to rank k % world, and the tail is completed by wrapping round to the start of the
            # permutation so that every rank runs the same number of full batches
            unit = self.bs * self.world
            total = (len(idx) + unit - 1) // unit * unit
            idx = (idx * (total // len(idx) + 1))[:total]
        batches = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        return batches[self.rank::self.world]

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        batches = self._batches()
        self.epoch += 1

        def make(ix):
            return collate_u8([self.ds[i] for i in ix], self.tok, self.res, self.pep)

        if self.prefetch == 0:
            for ix in batches:
                yield make(ix)
            return
        q: Queue = Queue(maxsize=self.prefetch)

        def worker():
            try:
                for ix in batches:
                    q.put(make(ix))
                q.put(None)
            except BaseException as e:  # surface decoding errors in the consumer
                q.put(e)
        threading.Thread(target=worker, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
