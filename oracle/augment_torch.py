"""CPU oracle for the trainer's augmentations: torch restatement of the torchvision functional ops the reference calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: torchvision (poetry.lock pins 0.15.2) is not installed in this
image and the reference has no augmentation tests; this file restates the published algorithms of
``torchvision.transforms._functional_tensor`` (``_blend``, ``rgb_to_grayscale``, ``adjust_brightness / contrast / saturation / hue``,
``_rgb2hsv``, ``_hsv2rgb``) and of ``torch.nn.functional.pad(mode="reflect")`` + ``crop`` exactly as the reference chains them in
``augment_data`` (diffusion/train_controlnet_genima.py:775-830).  NCHW float tensors, values in [0, 1] for the colour ops.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _blend(a: Tensor, b: Tensor, ratio: float) -> Tensor:
    return (ratio * a + (1.0 - ratio) * b).clamp(0.0, 1.0)


def rgb_to_grayscale(img: Tensor) -> Tensor:
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def adjust_brightness(img: Tensor, f: float) -> Tensor:
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img: Tensor, f: float) -> Tensor:
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def adjust_saturation(img: Tensor, f: float) -> Tensor:
    return _blend(img, rgb_to_grayscale(img), f)


def _rgb2hsv(img: Tensor) -> Tensor:
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cr_divisor, (maxc - g) / cr_divisor, (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img: Tensor) -> Tensor:
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img: Tensor, hue_factor: float) -> Tensor:
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter(img: Tensor, order, factors) -> Tensor:
    """ColorJitter.forward with the drawn (fn_idx order, factors by op id)."""
    for op in order:
        img = _OPS[op](img, factors[op])
    return img


def reflect_pad_crop(img: Tensor, i: int, j: int, pad: int = 2) -> Tensor:
    H, W = img.shape[-2:]
    return F.pad(img, (pad, pad, pad, pad), mode="reflect")[..., i:i + H, j:j + W]
