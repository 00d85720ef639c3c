"""Seeded synthetic weights + checkpoint IO for the Genima hot path.

No checkpoints exist on the build or GPU boxes (no network), so benchmarks and parity tests use weights
drawn from an in-repo *counter-based* PRNG: element ``i`` of tensor ``name`` depends only on
``(seed, name, i)``, never on numpy/torch generator state or version, so the CPU oracle and the HIP
path see bit-identical fp32 master weights on every box.  Scales follow SURVEY.md §8(d): conv/linear
~ U(-a, a) with std 1/sqrt(fan_in), norm gamma = 1 +- 0.1, beta = +-0.1, non-zero ControlNet
"zero convs" (std 0.02) so the residual path is exercised.

Real checkpoints: ``load_diffusers_dir`` reads the layout the reference loads
(``config.json`` + ``diffusion_pytorch_model[.fp16].safetensors`` /``model[.fp16].safetensors``;
controller/agent/sd_controlnet_agent.py:21-42) and ``save_diffusers_dir`` writes it
(diffusion/train_controlnet_genima.py:1077-1105, :1486).
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def counter_uniform(seed: int, name: str, n: int) -> np.ndarray:
    """``n`` float32 values uniform in [-1, 1), a pure function of (seed, name, index)."""
    base = _splitmix64(np.array([(seed * 0x9E3779B97F4A7C15 + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF],
                                dtype=np.uint64))[0]
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        with np.errstate(over="ignore"):
            idx = (np.arange(s, e, dtype=np.uint64) + base) & _M64
        z = _splitmix64(idx)
        u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))  # [0,1), 24 bits
        out[s:e] = u * np.float32(2.0) - np.float32(1.0)
    return out


def _s64(x: int) -> int:
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    return (x >> k) & ((1 << (64 - k)) - 1)  # logical shift on two's-complement int64


def _splitmix64_t(x: torch.Tensor) -> torch.Tensor:
    x = x + _s64(0x9E3779B97F4A7C15)
    z = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def counter_uniform_torch(seed: int, name: str, n: int, device) -> torch.Tensor:
    """Bit-identical to ``counter_uniform`` (tests/test_cpu_oracle.py) but runs as int64 torch ops on ``device`` -- the
    1.65 G synthetic parameters of the full SD-Turbo family are drawn on the GPU in seconds instead of minutes of numpy."""
    b0 = torch.tensor([_s64(seed * 0x9E3779B97F4A7C15 + _fnv1a64(name))], dtype=torch.int64, device=device)
    base = _splitmix64_t(b0)
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = 1 << 26
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        z = _splitmix64_t(torch.arange(s, e, dtype=torch.int64, device=device) + base)
        out[s:e] = _lsr(z, 40).to(torch.float32) * (1.0 / (1 << 24)) * 2.0 - 1.0
    return out


def counter_bytes(seed: int, name: str, n: int) -> np.ndarray:
    """``n`` uint8 values (synthetic images)."""
    u = counter_uniform(seed, name, n)
    return np.clip(np.floor((u * 0.5 + 0.5) * 256.0), 0, 255).astype(np.uint8)


_SQRT3 = 3.0 ** 0.5


def _init_rule(name: str, shape):
    """Return (kind, scale) for a parameter."""
    leaf = name.rsplit(".", 1)[-1]
    is_norm = any(t in name for t in (".norm", "norm_out", "layer_norm", "group_norm", "ln_", ".bn", "downsample.1."))
    if is_norm and len(shape) == 1:
        return ("gamma", 0.1) if leaf in ("weight", "running_var") else ("beta", 0.1)
    if leaf == "bias":
        return ("uniform", 0.02 * _SQRT3)
    if "token_embedding" in name or "position_embedding" in name or name.endswith("embed.weight"):
        return ("uniform", 0.02 * _SQRT3)
    if name.startswith("controlnet_down_blocks") or name.startswith("controlnet_mid_block") \
            or name.startswith("controlnet_cond_embedding.conv_out"):
        return ("uniform", 0.02 * _SQRT3)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return ("uniform", _SQRT3 / max(fan_in, 1) ** 0.5)


def synth_state_dict(schema: "OrderedDict[str, tuple]", seed: int, prefix: str = "", device="cpu") -> "OrderedDict[str, torch.Tensor]":
    """fp32 master state dict for ``schema`` (see module docstring).  ``device='cuda'`` draws on the GPU (same bits)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    on_cpu = torch.device(device).type == "cpu"
    for name, shape in schema.items():
        n = int(np.prod(shape)) if len(shape) else 1
        kind, scale = _init_rule(name, shape)
        if on_cpu:
            u = torch.from_numpy(counter_uniform(seed, prefix + name, n))
        else:
            u = counter_uniform_torch(seed, prefix + name, n, device)
        # identical f32 op order on both paths: scale * u (+ 1.0)
        v = u * np.float32(scale)
        if kind == "gamma":
            v = v + 1.0
        sd[name] = v.reshape(shape)
    return sd


def round_to(sd: Dict[str, torch.Tensor], dtype: torch.dtype) -> "OrderedDict[str, torch.Tensor]":
    """Round master weights through ``dtype`` and back to fp32 (what an fp16 checkpoint holds)."""
    return OrderedDict((k, v.to(dtype).to(torch.float32)) for k, v in sd.items())


# ----------------------------------------------------------------------------- diffusers-style directories
_WEIGHT_STEMS = ("diffusion_pytorch_model", "model")


def _load_safetensors(d: str, stem: str, variant: Optional[str]):
    """One component's tensors from ``<stem>[.<variant>].safetensors`` or its sharded form (``<stem>[.<variant>].safetensors.index.json``
    + ``<stem>-0000x-of-0000y[.<variant>].safetensors``); None if neither exists."""
    from safetensors.torch import load_file

    suffix = f".{variant}" if variant else ""
    single = os.path.join(d, f"{stem}{suffix}.safetensors")
    if os.path.exists(single):
        return load_file(single)
    index = os.path.join(d, f"{stem}.safetensors.index{suffix}.json")  # diffusers' naming; transformers puts the variant first
    if not os.path.exists(index):
        index = os.path.join(d, f"{stem}{suffix}.safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        sd = {}
        for fn in shards:
            sd.update(load_file(os.path.join(d, fn)))
        return sd
    return None


def load_diffusers_dir(path: str, subfolder: Optional[str] = None, variant: Optional[str] = None):
    """Return (config dict, fp32 state dict) from a diffusers/transformers component directory.  ``variant=None`` reads the
    full-precision file and ``variant="fp16"`` the ``.fp16`` one, as diffusers does; when only the other one exists it is used
    (diffusers would raise or warn depending on the version -- a directory saved by this repo always holds the plain name)."""
    d = os.path.join(path, subfolder) if subfolder else path
    with open(os.path.join(d, "config.json")) as f:
        cfg = json.load(f)
    order = [variant, None] if variant else [None, "fp16"]
    for v in order:
        for stem in _WEIGHT_STEMS:
            sd = _load_safetensors(d, stem, v)
            if sd is not None:
                return cfg, OrderedDict((k, t.to(torch.float32)) for k, t in sd.items())
    raise FileNotFoundError(f"no safetensors weight file under {d} (looked for {_WEIGHT_STEMS} with variants {order}, single or sharded)")


def save_diffusers_dir(path: str, cfg: dict, sd: Dict[str, torch.Tensor], dtype=torch.float32,
                       weight_name: str = "diffusion_pytorch_model.safetensors"):
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    save_file({k: v.detach().to(dtype).contiguous().cpu() for k, v in sd.items()}, os.path.join(path, weight_name))
