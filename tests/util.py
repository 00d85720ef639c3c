"""Shared helpers for the parity tests."""
import numpy as np
import torch

# The parity bar (BASELINE.json north_star): 1e-3 relative at fp16 tolerance.  Kernel outputs are compared with the fp32
# oracle evaluated on the SAME f16-rounded inputs: relative L2 error <= REL_L2 and every element within
# ABS_FRAC * max|ref| + one f16 ulp of the reference value.
REL_L2 = 1e-3


def rel_l2(y: torch.Tensor, ref: torch.Tensor) -> float:
    y, ref = y.detach().double().cpu(), ref.detach().double().cpu()
    return float((y - ref).norm() / ref.norm().clamp_min(1e-30))


def assert_close(y, ref, rel=REL_L2, what=""):
    y32, r32 = y.detach().float().cpu(), ref.detach().float().cpu()
    assert y32.shape == r32.shape, (what, y32.shape, r32.shape)
    assert torch.isfinite(y32).all(), f"{what}: non-finite output"
    e = rel_l2(y32, r32)
    mx = float((y32 - r32).abs().max())
    scale = float(r32.abs().max())
    assert e <= rel, f"{what}: rel L2 {e:.3e} > {rel:.1e} (max abs err {mx:.3e}, max |ref| {scale:.3e})"
    # elementwise: within f16 rounding of the reference plus accumulated-order noise
    tol = 2e-3 * scale + 1e-3
    assert mx <= tol, f"{what}: max abs err {mx:.3e} > {tol:.3e}"
    return e


def randn_h(*shape, seed=0, scale=1.0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16).to(device)


def q16(t):
    return t.half().float()
