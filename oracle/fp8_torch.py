"""CPU restatement of the fp8 (OCP e4m3) Linear -- TEST INFRASTRUCTURE ONLY (tests/, smoke, the bench's CPU leg); the product
never imports it.  Parity unpinned against the reference: the reference runs its SDXL Linears under fp16 autocast
(diffusion/train_controlnet_sdxl_genima.py:1448-1471) and has no fp8 path; BASELINE.json configs[4] asks for fp8 MFMA, so the
oracle pins the *arithmetic* of this repo's scheme instead: row-wise dynamic scaling, torch's float8_e4m3fn round-to-nearest-even
cast, exact products, f32 accumulation, dequantisation by scale_a[m] * scale_w[n], then the f16 Linear's epilogue.
"""
import torch


def quantize_rows(x: torch.Tensor):
    """x [rows, K] (any float dtype) -> (q float8_e4m3fn [rows, K], scale f32 [rows]); mirrors gn_quantize_fp8_rows exactly:
    amax over the row in f32, scale = amax / 448 (1 for a zero row), q = rne(x * (448 / amax))."""
    x = x.float()
    amax = x.abs().amax(dim=-1)
    pos = amax > 0
    c448 = torch.full_like(amax, 448.0)
    scale = torch.where(pos, amax / c448, torch.ones_like(amax))
    # tensor / tensor: IEEE division.  (`448.0 / amax` is reciprocal-then-multiply in torch -- two roundings -- and moves values
    # across e4m3 rounding ties about once per 10^4 elements.)
    inv = torch.where(pos, c448 / amax.clamp_min(1e-30), torch.zeros_like(amax))
    q = (x * inv[:, None]).to(torch.float8_e4m3fn)
    return q, scale


def linear_fp8(x: torch.Tensor, w: torch.Tensor, bias=None, act=None, residual=None) -> torch.Tensor:
    """x [M, K], w [N, K] (f16-representable values) -> f32 [M, N]."""
    xq, xs = quantize_rows(x)
    wq, ws = quantize_rows(w)
    y = (xq.float().double() @ wq.float().double().t()).float() * (xs[:, None] * ws[None, :])
    if bias is not None:
        y = y + bias.float()
    if act is not None:
        y = act(y)
    if residual is not None:
        y = y + residual.float()
    return y
