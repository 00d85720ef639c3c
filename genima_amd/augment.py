"""Train-time augmentation on the device: the reference's ``augment_data`` (diffusion/train_controlnet_genima.py:775-830) for the
README recipe ``--augmentations=crop,colorjitter`` (README.md:204), on NHWC f16 8-channel batches already resident in HBM.

The random draws follow torchvision's order with the torch CPU generator, so a run seeded like the reference's draws the same
jitter factors / op order / crop offsets:
  ColorJitter.get_params: ``fn_idx = randperm(4)``; then brightness, contrast, saturation, hue factors, each
  ``float(torch.empty(1).uniform_(lo, hi))`` -- ONE draw for the whole batch tensor (the reference calls the transform on the batch);
  RandomCrop.get_params on the reflect-padded image: ``i = randint(0, h - th + 1)``, ``j = randint(0, w - tw + 1)``.
``elastic`` / ``blur`` / ``affine`` (not in the README recipe) are not built and raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from ._lib import check
from .engine import Engine, _ptr

JITTER = dict(brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.9, 1.1), hue=(-0.05, 0.05))  # ColorJitter(0.2, 0.2, 0.1, 0.05)
CROP_PAD = 2


def draw_color_jitter(generator: Optional[torch.Generator] = None) -> Tuple[Tuple[int, ...], Tuple[float, ...]]:
    """-> (op order, factors indexed by op id 0 brightness / 1 contrast / 2 saturation / 3 hue), torchvision's draw order."""
    order = tuple(int(v) for v in torch.randperm(4, generator=generator))
    factors = tuple(float(torch.empty(1).uniform_(lo, hi, generator=generator)) for lo, hi in
                    (JITTER["brightness"], JITTER["contrast"], JITTER["saturation"], JITTER["hue"]))
    return order, factors


def draw_crop(pad: int = CROP_PAD, generator: Optional[torch.Generator] = None) -> Tuple[int, int]:
    i = int(torch.randint(0, 2 * pad + 1, size=(1,), generator=generator))
    j = int(torch.randint(0, 2 * pad + 1, size=(1,), generator=generator))
    return i, j


def color_jitter(E: Engine, x: torch.Tensor, order, factors, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: f16 [B, H, W, ld >= 3], RGB in [0, 1]."""
    B, H, W, ld = x.shape
    out = torch.empty_like(x) if out is None else out
    ws = E._workspace(int(E.lib.gn_color_jitter_workspace_bytes(B)))
    check(E.lib.gn_color_jitter(E._ctx, _ptr(x), _ptr(out), B, H * W, ld, (C.c_int32 * 4)(*order), (C.c_float * 4)(*factors), _ptr(ws)), "gn_color_jitter")
    return out


def reflect_pad_crop(E: Engine, x: torch.Tensor, i: int, j: int, pad: int = CROP_PAD) -> torch.Tensor:
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    check(E.lib.gn_reflect_pad_crop(E._ctx, _ptr(x), _ptr(out), B, H, W, Cc, pad, i, j), "gn_reflect_pad_crop")
    return out


def augment_data(E: Engine, augmentations: Optional[str], batch: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None):
    """``augment_data(args, batch)`` with ``args.augmentations`` given as the comma list; batch tensors NHWC f16 8-channel on the device."""
    images, cond = batch["pixel_values"], batch["conditioning_pixel_values"]
    if augmentations:
        augs = [a for a in augmentations.split(",") if a]
        unsupported = [a for a in augs if a not in ("colorjitter", "crop")]
        if unsupported:
            raise NotImplementedError(f"augmentations {unsupported} are not built on the HIP path (README recipe: crop,colorjitter)")
        if "colorjitter" in augs:
            cond = color_jitter(E, cond, *draw_color_jitter(generator))
        if "crop" in augs:
            i, j = draw_crop(CROP_PAD, generator)
            images = reflect_pad_crop(E, images, i, j)
            cond = reflect_pad_crop(E, cond, i, j)
    out = dict(batch)
    out["pixel_values"], out["conditioning_pixel_values"] = images, cond
    return out
