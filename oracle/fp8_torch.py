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


# ---- fp8 attention (csrc/attention_fp8.hip) ----------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634


def attention_operands(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float):
    """q, k, v [B, N, heads*64] (f16-representable) -> (q8, k8 [B, N, C], v8t [B, C, Npad]) uint8, what gn_attention_fp8_quantize
    writes: q8 = e4m3(q * (scale * log2 e)), k8 = e4m3(k), saturating at +-448; v8t = e4m3(v) transposed with the keys of every
    64-key tile in the MFMA operand order (position 32*hi + 16*u + 4*g + i holds key 32*u + 8*g + 4*hi + i), keys >= N zero."""
    B, N, C = q.shape
    Np = (N + 63) // 64 * 64
    qs = torch.tensor(scale, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)  # the kernel's f32 product
    q8 = (q.float() * qs).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    k8 = k.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    v8 = torch.zeros(B, Np, C, dtype=torch.uint8)
    v8[:, :N] = v.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    pos = torch.arange(64)
    hi, u, g, i = pos >> 5, (pos >> 4) & 1, (pos >> 2) & 3, pos & 3
    key_of_pos = 32 * u + 8 * g + 4 * hi + i
    v8 = v8.view(B, Np // 64, 64, C)[:, :, key_of_pos].reshape(B, Np, C)
    return q8, k8, v8.transpose(1, 2).contiguous()


def attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int):
    """The attention the fp8 kernel approximates: EXACT e4m3 operands (as above), f64 softmax and products.  Returns (o f32
    [B, N, C], lse f32 [B, heads, N] in the log2 domain of the scaled scores).  What the kernel adds on top is the e4m3 rounding of
    the probabilities (3 mantissa bits: <= 6.25 % per element, zero-mean) and the f16 rounding of o."""
    B, N, C = q.shape
    D = C // heads
    deq = lambda t: t.view(torch.float8_e4m3fn).double()
    qs = torch.tensor(D ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)
    qd = deq((q.float() * qs).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)).view(B, N, heads, D).transpose(1, 2)
    kd = deq(k.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)).view(B, N, heads, D).transpose(1, 2)
    vd = deq(v.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)).view(B, N, heads, D).transpose(1, 2)
    s = qd @ kd.transpose(-1, -2)  # exponent units (log2 domain)
    m = s.amax(dim=-1, keepdim=True)
    p = torch.exp2(s - m)
    l = p.sum(dim=-1, keepdim=True)
    o = (p @ vd) / l
    lse = (m + torch.log2(l)).squeeze(-1)
    return o.transpose(1, 2).reshape(B, N, C).float(), lse.float()
