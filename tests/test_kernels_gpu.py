"""-m gpu: per-kernel parity of libgenima_hip.so (through the C ABI via genima_amd.engine) against fp32 torch-CPU
restatements of the same op on the same f16-rounded inputs (the oracle for floating-point kernels)."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import assert_close, q16, randn_h, rel_l2

pytestmark = pytest.mark.gpu

ACT = {"none": 0, "silu": 1, "gelu": 2, "quick_gelu": 3, "relu": 4}


def ref_act(x, act):
    return {"none": lambda v: v, "silu": F.silu, "gelu": F.gelu, "quick_gelu": lambda v: v * torch.sigmoid(1.702 * v),
            "relu": F.relu}[act](x)


# ---------------------------------------------------------------------------------------------------- GEMM (Linear)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 320, 320), (77, 1024, 1024), (4096, 640, 2560), (64, 1280, 5120),
                                   (8, 2560, 1280), (130, 192, 72)])
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_linear(engine, M, N, K, act):
    x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3, scale=0.1)
    r = randn_h(M, N, seed=4)
    y = engine.linear(x, w, b, act=ACT[act], residual=r)
    ref = ref_act(x.float().cpu() @ w.float().cpu().t() + b.float().cpu(), act) + r.float().cpu()
    assert_close(y, ref, what=f"linear {M}x{N}x{K} {act}")


def test_linear_transposed_mfma_layout(engine):
    """A = I with an ASYMMETRIC W catches a row/col swap in the MFMA C/D mapping (cdna_hip_programming.md rule 16)."""
    K = N = 128
    x = torch.eye(K, dtype=torch.float16, device="cuda")
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 / 64.0).to(torch.float16).cuda()
    y = engine.linear(x, w)
    assert torch.equal(y.float().cpu(), w.float().cpu().t())


@pytest.mark.parametrize("splitk", [2, 5])
def test_linear_splitk(engine, splitk):
    M, N, K = 96, 320, 2560
    x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3)
    y = engine.linear(x, w, b, act=ACT["silu"], splitk=splitk)
    ref = F.silu(x.float().cpu() @ w.float().cpu().t() + b.float().cpu())
    assert_close(y, ref, what=f"splitk {splitk}")


def test_linear_auto_splitk_small_m(engine):
    M, N, K = 64, 1280, 11520
    x, w = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5)
    y = engine.linear(x, w)
    assert_close(y, x.float().cpu() @ w.float().cpu().t(), what="auto splitk")


def test_geglu(engine):
    from genima_amd.packing import pack_geglu

    M, C = 200, 320
    x = randn_h(M, C, seed=1)
    w = torch.randn(8 * C, C, generator=torch.Generator().manual_seed(2)) * C ** -0.5
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    wp, bp = pack_geglu(q16(w), q16(b))
    y = engine.linear(x, wp.cuda(), bp.cuda(), act=5)
    h = x.float().cpu() @ q16(w).t() + q16(b)
    hid, gate = h.chunk(2, dim=-1)
    assert y.shape == (M, 4 * C)
    assert_close(y, hid * F.gelu(gate), what="geglu")


def test_linear_transposed_out(engine):
    B, L, K, N = 3, 77, 128, 192
    x, w = randn_h(B, L, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5)
    vt = engine.linear(x, w, transposed_out=True, rows_per_batch=L, pad_cols=128)
    ref = (x.float().cpu() @ w.float().cpu().t()).transpose(1, 2)
    assert vt.shape == (B, N, 128)
    assert_close(vt[:, :, :L], ref, what="transposed out")
    assert float(vt[:, :, L:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------- conv
def pack(w):
    from genima_amd.packing import pack_conv_weight

    return pack_conv_weight(w).cuda()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride", [
    (2, 64, 64, 16, 16, 3, 1), (1, 320, 320, 32, 32, 3, 1), (2, 128, 256, 17, 23, 3, 2), (1, 640, 320, 8, 8, 1, 1),
    (1, 8, 320, 64, 64, 3, 1), (2, 16, 32, 40, 40, 3, 2), (1, 96, 96, 24, 24, 3, 1), (1, 320, 8, 32, 32, 3, 1),
    (1, 8, 64, 64, 64, 7, 2), (1, 1280, 1280, 8, 8, 3, 1)])
def test_conv2d(engine, B, Cin, Cout, H, W, k, stride):
    g = torch.Generator().manual_seed(5)
    x = q16(torch.randn(B, Cin, H, W, generator=g))
    w = q16(torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5)
    b = q16(torch.randn(Cout, generator=g) * 0.1)
    y = engine.conv2d(nhwc(x).half().cuda(), pack(w), b.half().cuda(), ksize=k, stride=stride)
    ref = F.conv2d(x, w, b, stride=stride, padding=k // 2)
    assert_close(y, nhwc(ref), what=f"conv {Cin}->{Cout} {H}x{W} k{k} s{stride}")


def test_conv2d_fused_epilogue_concat_upsample(engine):
    """Virtual concat + nearest-2x upsample folded into the gather + time shift + SiLU + residual in one launch."""
    g = torch.Generator().manual_seed(6)
    B, C1, C2, Cout, H, W = 2, 64, 128, 192, 12, 10
    x1, x2 = q16(torch.randn(B, C1, H, W, generator=g)), q16(torch.randn(B, C2, H, W, generator=g))
    w = q16(torch.randn(Cout, C1 + C2, 3, 3, generator=g) * (9 * (C1 + C2)) ** -0.5)
    b = q16(torch.randn(Cout, generator=g) * 0.1)
    sh = q16(torch.randn(B, Cout, generator=g))
    res = q16(torch.randn(B, Cout, 2 * H, 2 * W, generator=g))
    y = engine.conv2d(nhwc(x1).half().cuda(), pack(w), b.half().cuda(), x2=nhwc(x2).half().cuda(), shift=sh.half().cuda(),
                      residual=nhwc(res).half().cuda(), act=1, upsample2x=True)
    up = F.interpolate(torch.cat([x1, x2], 1), scale_factor=2.0, mode="nearest")
    ref = F.silu(F.conv2d(up, w, b, padding=1) + sh[:, :, None, None]) + res
    assert_close(y, nhwc(ref), what="fused conv")


def test_conv2d_asymmetric_pad(engine):
    """VAE encoder downsample: F.pad(0,1,0,1) then 3x3 stride 2 pad 0."""
    g = torch.Generator().manual_seed(7)
    x = q16(torch.randn(1, 64, 16, 16, generator=g))
    w = q16(torch.randn(64, 64, 3, 3, generator=g) / 24.0)
    y = engine.conv2d(nhwc(x).half().cuda(), pack(w), None, stride=2, pad=(0, 0, 1, 1))
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    assert_close(y, nhwc(ref), what="asym pad conv")


# ---------------------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, heads, causal=False, f16_storage=False):
    """fp32 attention of the f16-rounded inputs; f16_storage: what ANY f16-storage kernel computes -- the probabilities (relative to the
    row maximum) rounded to f16 before P.V, the row sum taken of those rounded values, the output rounded to f16."""
    B, Nq, C = q.shape
    d = C // heads
    qh, kh, vh = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * d ** -0.5
    if causal:
        s = s + torch.full((Nq, k.shape[1]), float("-inf")).triu(1)
    if f16_storage:
        pr = q16(torch.exp(s - s.amax(-1, keepdim=True)))
        return q16((pr @ vh) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, Nq, C)
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Nq, C)


def assert_attention(o, q, k, v, heads, causal=False, what="", factor=1.35):
    """The attention bar.  BASELINE north_star's 1e-3 (relative, fp16 tolerance) is asserted OUTRIGHT: rel-L2 from the fp32 softmax(QK^T)V of
    the same f16 inputs < 1e-3.  On top of it the kernel may not be worse than `factor` x what ANY f16-storage attention costs (the
    restatement above: P and O rounded to f16), + 5e-5.  Measured on MI355X (printed with -s): 2.7e-4 .. 3.6e-4 against 2.3e-4 .. 2.8e-4 for
    the restatement = 1.21 .. 1.29x on all plain shapes (the kernel also rounds Q * scale * log2 e to f16: the MFMA then yields exponents);
    rows that re-reference their softmax mid-sequence (spiked keys, the stream kernel's fallback) reach 1.7 .. 2.1x, still <= 4.1e-4."""
    q, k, v = (t.float().cpu() for t in (q, k, v))
    r32, r16 = ref_attention(q, k, v, heads, causal), ref_attention(q, k, v, heads, causal, f16_storage=True)
    e, e16 = rel_l2(o.float().cpu(), r32), rel_l2(r16, r32)
    print(f"attention {what}: rel-L2 {e:.3e} (f16-storage restatement {e16:.3e}, ratio {e / max(e16, 1e-12):.2f})")
    assert_close(o, r32, rel=1e-3, what=what)  # shape, finiteness, rel-L2 < 1e-3, the elementwise bound
    assert e <= factor * e16 + 5e-5, f"{what}: {e:.3e} vs f16-storage {e16:.3e}"
    return e


@pytest.mark.parametrize("B,heads,Nq,Nk,D,causal", [
    (2, 5, 256, 256, 64, False), (1, 2, 1024, 1024, 64, False), (2, 4, 64, 77, 64, False), (2, 16, 77, 77, 64, True),
    (1, 10, 300, 77, 64, False), (2, 8, 259, 259, 32, False), (1, 8, 20, 259, 32, False), (1, 1, 4096, 4096, 64, False)])
def test_attention(engine, B, heads, Nq, Nk, D, causal):
    C = heads * D
    q, k, v = randn_h(B, Nq, C, seed=1), randn_h(B, Nk, C, seed=2), randn_h(B, Nk, C, seed=3)
    Np = (Nk + 63) // 64 * 64
    vt = torch.full((B, C, Np), float("nan"), dtype=torch.float16, device="cuda")  # pad columns are never trusted
    vt[:, :, :Nk] = v.transpose(1, 2)
    o = engine.attention(q, k, vt, heads, Nk=Nk, causal=causal)
    assert_attention(o, q, k, v, heads, causal, what=f"{B}x{heads}x{Nq}x{Nk}x{D} causal={causal}")


def test_attention_strided_qk_and_spike(engine):
    """q|k as column slices of one fused projection, and a spiked key forcing a large online-softmax rescale mid-sequence."""
    B, heads, N, D = 1, 5, 512, 64
    C = heads * D
    qk = randn_h(B, N, 2 * C, seed=1)
    qk[0, 300, C:] *= 6.0
    v = randn_h(B, N, C, seed=3)
    vt = v.transpose(1, 2).contiguous()
    o = engine.attention(qk[:, :, :C], qk[:, :, C:], vt, heads)
    assert_attention(o, qk[:, :, :C], qk[:, :, C:], v, heads, what="strided/spiked", factor=2.0)


@pytest.mark.parametrize("where", ["far_tile", "one_row", "every_tile"])
def test_attention_stream_kernel_fallback(engine, attn_variant, where):
    """attention_stream.hip guesses the softmax reference from the block's diagonal key tile and never looks back inside the loop; scores
    that outgrow the guess by more than 2^8 set a flag and the block redoes its rows with the max-tracking loop.  Force that: keys far
    from the diagonal (or everywhere, growing) that beat every diagonal score by tens of nats."""
    B, heads, N, D = 2, 3, 1024, 64
    C = heads * D
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, N, C, generator=g) for _ in range(3))
    if where == "far_tile":      # one key in tile 9 aligned with EVERY query of head 0: score ~ +40 nats
        d = torch.randn(D, generator=g); d = d / d.norm()
        q[:, :, :D] += 6.0 * d
        k[:, 600, :D] = 50.0 * d
    elif where == "one_row":     # a single query row (one lane pair of one wave) with one huge key
        k[0, 37, D:2 * D] = 12.0 * q[0, 900, D:2 * D]
    else:                        # the maximum keeps growing along the keys for all rows of head 2
        d = torch.randn(D, generator=g); d = d / d.norm()
        q[:, :, 2 * D:] = 0.3 * q[:, :, 2 * D:] + 8.0 * d
        k[:, :, 2 * D:] = 0.3 * k[:, :, 2 * D:] + torch.linspace(-4.0, 4.0, N)[None, :, None] * d * 3.0
    q, k, v = (t.half().cuda() for t in (q, k, v))
    vt = v.transpose(1, 2).contiguous()
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    attn_variant(4)
    o = engine.attention(q, k, vt, heads, lse=lse)
    assert_attention(o, q, k, v, heads, what=f"stream-kernel fallback ({where})", factor=2.5)
    s = (q.float().view(B, N, heads, D).transpose(1, 2) @ k.float().view(B, N, heads, D).transpose(1, 2).transpose(-1, -2)) * D ** -0.5
    ref_lse = (torch.logsumexp(s, -1) * 1.4426950408889634).cpu()
    assert float((lse.cpu() - ref_lse).abs().max()) < 2e-2, "lse after the fallback"
    o2 = engine.attention(q, k, v, heads, v_rowmajor=True)  # the generic kernel on the same problem
    assert_close(o, o2.float(), rel=1e-3, what=f"stream kernel vs generic kernel ({where})")


@pytest.fixture
def attn_variant(engine):
    """gn_attention_set_variant for the duration of a test (0 attention.hip, 4 attention_stream.hip, 5 attention_pwg.hip; -1 the library's choice)."""
    yield engine.lib.gn_attention_set_variant
    engine.lib.gn_attention_set_variant(-1)


@pytest.mark.parametrize("B,heads,N", [(1, 2, 256), (2, 3, 512), (1, 2, 1024), (1, 2, 320), (1, 1, 128), (1, 3, 4096), (3, 11, 2048), (2, 65, 1024)])
def test_attention_pwg_kernel(engine, attn_variant, B, heads, N):
    """attention_pwg.hip (one wave per SIMD, 64 query rows per wave): 256-row blocks (2 x 65 x 1024: 520 blocks = two full rounds + 8 -> 16
    split blocks), all-split grids (block count <= 128 with >= 8 key tiles: 512 .. 4096 keys), short key loops (128, 256 keys) and ragged
    row counts (320: masked rows, no split) -- against fp32 at the attention bar, against attention_stream.hip, and the lse output."""
    C = heads * 64
    qk, v = randn_h(B, N, 2 * C, seed=41), randn_h(B, N, C, seed=43)
    q, k = qk[:, :, :C], qk[:, :, C:]  # column slices of one fused projection
    vt = v.transpose(1, 2).contiguous()
    attn_variant(5)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    o = engine.attention(q, k, vt, heads, lse=lse).clone()
    attn_variant(4)
    o4 = engine.attention(q, k, vt, heads).clone()
    assert_attention(o, q, k, v, heads, what=f"pwg {B}x{heads}x{N}")
    assert_close(o, o4.float(), rel=1e-3, what="attention_pwg vs attention_stream")
    s = (q.float().cpu().view(B, N, heads, 64).transpose(1, 2) @ k.float().cpu().view(B, N, heads, 64).transpose(1, 2).transpose(-1, -2)) * 0.125
    assert float((lse.cpu() - torch.logsumexp(s, -1) * 1.4426950408889634).abs().max()) < 2e-3, "lse (log2 units)"
    attn_variant(5)
    assert torch.equal(o, engine.attention(q, k, vt, heads)), "a second call is bit-identical"
    if B > 1:  # which rows run in split blocks depends on the head, never on the batch position
        perm = torch.arange(B - 1, -1, -1, device="cuda")
        o_p = engine.attention(qk[perm][:, :, :C], qk[perm][:, :, C:], vt[perm].contiguous(), heads)
        assert torch.equal(o_p, o[perm]), "permuting the batch permutes the output bit for bit"


def test_attention_default_routing_takes_the_pwg_kernel_for_large_grids(engine, attn_variant):
    """gn_attention_fwd's own choice: >= 2048 keys and >= 256 row blocks of 256 (the 64 x 64 latent level at B >= 4) run attention_pwg.hip --
    bit-identical to variant 5 -- smaller grids and shorter key sets stay with attention_stream.hip (variant 4)."""
    def run(B, heads, N, variant):
        C = heads * 64
        qk, v = randn_h(B, N, 2 * C, seed=51), randn_h(B, N, C, seed=52)
        attn_variant(variant)
        return engine.attention(qk[:, :, :C], qk[:, :, C:], v.transpose(1, 2).contiguous(), heads).clone()
    big = (4, 5, 4096)   # 320 blocks: 256 full + 128 split
    assert torch.equal(run(*big, -1), run(*big, 5))
    small = (1, 5, 4096)  # 80 blocks: below the block threshold
    assert torch.equal(run(*small, -1), run(*small, 4))
    short = (8, 10, 1024)  # 320 blocks, 1024 keys: below the key threshold
    assert torch.equal(run(*short, -1), run(*short, 4))


@pytest.mark.parametrize("where,N", [("far_tile", 1024), ("one_row", 1024), ("every_tile", 1024), ("far_tile", 448), ("every_tile", 4096)])
def test_attention_pwg_kernel_fallback(engine, attn_variant, where, N):
    """The optimistic softmax's fallback in attention_pwg.hip, in split blocks (1024 / 4096 keys: both waves of a row block redo all keys)
    and in 256-row blocks (448 keys: seven tiles): the cases of test_attention_stream_kernel_fallback."""
    B, heads, D = 2, 3, 64
    C = heads * D
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, N, C, generator=g) for _ in range(3))
    if where == "far_tile":
        d = torch.randn(D, generator=g); d = d / d.norm()
        q[:, :, :D] += 6.0 * d
        k[:, N - 40, :D] = 50.0 * d
    elif where == "one_row":
        k[0, 37, D:2 * D] = 12.0 * q[0, 900, D:2 * D]
    else:
        d = torch.randn(D, generator=g); d = d / d.norm()
        q[:, :, 2 * D:] = 0.3 * q[:, :, 2 * D:] + 8.0 * d
        k[:, :, 2 * D:] = 0.3 * k[:, :, 2 * D:] + torch.linspace(-4.0, 4.0, N)[None, :, None] * d * 3.0
    q, k, v = (t.half().cuda() for t in (q, k, v))
    vt = v.transpose(1, 2).contiguous()
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    attn_variant(5)
    o = engine.attention(q, k, vt, heads, lse=lse)
    assert_attention(o, q, k, v, heads, what=f"pwg fallback ({where}, {N})", factor=2.5)
    s = (q.float().view(B, N, heads, D).transpose(1, 2) @ k.float().view(B, N, heads, D).transpose(1, 2).transpose(-1, -2)) * D ** -0.5
    ref_lse = (torch.logsumexp(s, -1) * 1.4426950408889634).cpu()
    assert float((lse.cpu() - ref_lse).abs().max()) < 2e-2, "lse after the fallback"
    o2 = engine.attention(q, k, v, heads, v_rowmajor=True)  # the generic kernel on the same problem
    assert_close(o, o2.float(), rel=1e-3, what=f"pwg kernel vs generic kernel ({where})")


@pytest.mark.parametrize("B,heads,Nq,Nk", [(2, 5, 4096, 77), (1, 10, 1000, 77), (2, 5, 16421, 77), (1, 3, 37, 65), (2, 2, 300, 96), (1, 20, 64, 80)])
def test_cross_attention_short_key_set(engine, B, heads, Nq, Nk):
    """The prompt cross-attention of every BasicTransformerBlock (64 < Nk <= 96 keys, D = 64, V^T given): ragged row counts, q as a column
    slice, NaN in V^T's pad columns, the lse output; against fp32 and against the row-major-V form of the same problem.  (Written for the
    register-resident key-set kernel of tools/probes/attn_cross_experiment.patch, which did not beat the generic kernel and is not built.)"""
    C = heads * 64
    qq, k, v = randn_h(B, Nq, C + 64, seed=31), randn_h(B, Nk, C, seed=32), randn_h(B, Nk, C, seed=33)
    q = qq[:, :, 64:]  # row stride C + 64
    vt = torch.full((B, C, 128), float("nan"), dtype=torch.float16, device="cuda")
    vt[:, :, :Nk] = v.transpose(1, 2)
    lse = torch.empty(B, heads, Nq, dtype=torch.float32, device="cuda")
    o = engine.attention(q, k, vt, heads, Nk=Nk, lse=lse)
    assert_attention(o, q, k, v, heads, what=f"cross {B}x{heads}x{Nq}x{Nk}")
    lse2 = torch.empty_like(lse)
    o2 = engine.attention(q, k, v, heads, Nk=Nk, v_rowmajor=True, lse=lse2)
    assert_close(o, o2.float(), rel=1e-3, what="V^T form vs row-major V")
    assert float((lse - lse2).abs().max()) < 2e-3, "lse (log2 units)"


@pytest.mark.parametrize("heads,Nq,Nk,causal", [(5, 512, 512, False), (4, 200, 77, False), (2, 77, 77, True)])
def test_attention_rowmajor_v(engine, heads, Nq, Nk, causal):
    """gn_attn_desc.v_rowmajor: V handed over row-major (a column slice of a q | k | v projection) and transposed out of the LDS tile
    by the kernel -- bit-identical to the V^T path, ragged key counts included."""
    C = heads * 64
    q, kv = randn_h(2, Nq, C, seed=21), randn_h(2, Nk, 2 * C, seed=22)
    k, v = kv[:, :, :C], kv[:, :, C:]
    pad = (Nk + 63) // 64 * 64
    vt = torch.zeros(2, C, pad, dtype=torch.float16, device="cuda")
    vt[:, :, :Nk] = v.transpose(1, 2)
    o1 = engine.attention(q, k, vt, heads, Nk=Nk, causal=causal).clone()
    o2 = engine.attention(q, k, v, heads, Nk=Nk, causal=causal, v_rowmajor=True)
    if causal or Nk % 64 or Nk < 128:
        assert torch.equal(o1, o2)  # one kernel, two ways to its V fragments
    else:  # the V^T form of these shapes runs the branch-free kernel (attention_stream.hip): another summation order
        assert_close(o1, o2.float(), rel=1e-3, what="V^T (stream kernel) vs row-major V (generic kernel)")
    assert_attention(o2, q, k, v, heads, causal, what="row-major V")
    # the block-shape overrides (GN_ATTN_VARIANT 1 / 2: tuning aids) have no row-major-V form: such a problem keeps its kernel (it used to be read as V^T,
    # far outside the tensor)
    for var in (1, 2):
        prev = engine.lib.gn_attention_set_variant(var)
        try:
            o3 = engine.attention(q, k, v, heads, Nk=Nk, causal=causal, v_rowmajor=True)
            engine.synchronize()
        finally:
            engine.lib.gn_attention_set_variant(prev)
        assert torch.equal(o3, o2), f"variant {var} with a row-major V"


# ---------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,H,W,C,act", [(2, 64, 64, 320, 1), (1, 8, 8, 1280, 1), (2, 32, 32, 128, 0), (1, 128, 128, 128, 1),
                                         (1, 16, 16, 2560, 1), (2, 7, 9, 64, 1)])
def test_groupnorm(engine, B, H, W, C, act):
    g = torch.Generator().manual_seed(8)
    x = q16(torch.randn(B, C, H, W, generator=g) * 2.0 + 0.5)
    gm, bt = q16(1 + 0.1 * torch.randn(C, generator=g)), q16(0.1 * torch.randn(C, generator=g))
    y = engine.groupnorm(nhwc(x).half().cuda(), gm.half().cuda(), bt.half().cuda(), 32, 1e-5, act=act)
    ref = F.group_norm(x, 32, gm, bt, 1e-5)
    ref = F.silu(ref) if act else ref
    assert_close(y, nhwc(ref), what=f"groupnorm {C}@{H}x{W}")


def test_groupnorm_concat(engine):
    g = torch.Generator().manual_seed(9)
    x1, x2 = q16(torch.randn(2, 320, 16, 16, generator=g)), q16(torch.randn(2, 640, 16, 16, generator=g) * 3)
    gm, bt = q16(1 + 0.1 * torch.randn(960, generator=g)), q16(0.1 * torch.randn(960, generator=g))
    y = engine.groupnorm(nhwc(x1).half().cuda(), gm.half().cuda(), bt.half().cuda(), 32, 1e-5, act=1, x2=nhwc(x2).half().cuda())
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, gm, bt, 1e-5))
    assert_close(y, nhwc(ref), what="groupnorm concat")


@pytest.mark.parametrize("M,C", [(4096, 320), (77, 1024), (1000, 1280), (5, 256), (130, 4096)])
def test_layernorm(engine, M, C):
    g = torch.Generator().manual_seed(10)
    x = q16(torch.randn(M, C, generator=g) * 2 + 1)
    gm, bt = q16(1 + 0.1 * torch.randn(C, generator=g)), q16(0.1 * torch.randn(C, generator=g))
    y = engine.layernorm(x.half().cuda(), gm.half().cuda(), bt.half().cuda(), 1e-5)
    assert_close(y, F.layer_norm(x, (C,), gm, bt, 1e-5), what=f"layernorm {M}x{C}")


# ---------------------------------------------------------------------------------------------------- small ops
def test_timestep_embedding(engine):
    from oracle.sd_torch import timestep_embedding

    t = torch.tensor([999.0, 799.0, 0.0, 199.0])
    y = engine.timestep_embedding(t.cuda(), 320)
    assert_close(y, timestep_embedding(t, 320), what="timestep embedding")


def test_euler_and_scale(engine):
    import numpy as np

    from oracle import scheduler as S

    g = torch.Generator().manual_seed(11)
    x = (torch.randn(2, 8, 8, 4, generator=g) * 14.6).half()
    eps8 = torch.randn(2, 8, 8, 8, generator=g).half()
    xs = engine.scale_pad(x.cuda(), 0.068265, 8)
    assert_close(xs[..., :4], x.float() * 0.068265, what="scale_pad")
    assert float(xs[..., 4:].abs().max()) == 0.0
    xd = x.clone().cuda()
    engine.euler_step(xd, eps8.cuda(), 14.614647, 5.087765)
    ref = S.euler_step(eps8[..., :4].numpy(), 14.614647, 5.087765, x.numpy())
    assert np.array_equal(xd.cpu().numpy(), ref), "euler step must match the oracle bit-for-bit in f16"


def test_image_pre_post(engine):
    from genima_amd.weights import counter_bytes
    from oracle.sd_torch import vae_postprocess_u8

    img = torch.from_numpy(counter_bytes(1, "img", 2 * 32 * 32 * 3).reshape(2, 32, 32, 3))
    f = engine.image_u8_to_f16(img.cuda(), 8)
    assert torch.equal(f[..., :3].cpu(), (img.float() / 255.0).half())
    assert float(f[..., 3:].abs().max()) == 0.0
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(2, 16, 16, 8, generator=g) * 0.8).half()
    u = engine.image_f16_to_u8(x.cuda())
    ref = vae_postprocess_u8(x[..., :3].permute(0, 3, 1, 2).float())
    diff = (u.cpu().int() - ref.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.02  # f16 (x/2+0.5) vs f32: rare 1-LSB ties


def test_misc_ops(engine):
    a, b = randn_h(4, 33, 64, seed=1), randn_h(4, 33, 64, seed=2)
    assert_close(engine.add(a, b), a.float().cpu() + b.float().cpu(), what="add")
    assert_close(engine.act(a, 1), F.silu(a.float().cpu()), what="silu")
    ids = torch.tensor([[5, 1, 0, 7], [2, 2, 9, 3]], dtype=torch.int32)
    tok, pos = randn_h(10, 64, seed=3), randn_h(4, 64, seed=4)
    e = engine.embedding(ids.cuda(), tok, pos)
    assert_close(e, tok.float().cpu()[ids.long()] + pos.float().cpu()[None], what="embedding")
    s = randn_h(37, 4096, seed=5, scale=3.0)
    ref = torch.softmax(s.float().cpu() * 0.25, -1)
    engine.softmax_rows(s, 0.25)
    assert_close(s, ref, what="softmax rows")
    x = randn_h(2, 17, 19, 64, seed=6)
    mp = engine.maxpool3x3s2(x)
    ref = F.max_pool2d(x.float().cpu().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(mp.float().cpu(), ref)


def test_pack_entry_points_match_the_python_packers(engine):
    """gn_pack_conv_weight / gn_pack_geglu_rows / gn_pack_fold_layernorm (the C-ABI repacking a non-Python host calls) against
    genima_amd/packing.py on the same tensors: the two layout shuffles bit for bit (f32 and f16 sources, channel counts that need
    padding), the LayerNorm fold's W * gamma bit for bit and its sums to f32 / f16 rounding."""
    import ctypes as C

    from genima_amd import packing
    from genima_amd.engine import check

    E = engine
    for (O, I, KH, KW) in ((20, 12, 3, 3), (64, 64, 1, 1), (8, 3, 7, 7), (136, 320, 3, 3)):
        w = torch.randn(O, I, KH, KW, device="cuda") * 0.1
        want = packing.pack_conv_weight(w)
        for src in (w, w.half()):
            got = torch.full_like(want, float("nan"))
            check(E.lib.gn_pack_conv_weight(E._ctx, C.c_void_p(src.data_ptr()), int(src.dtype == torch.float16), C.c_void_p(got.data_ptr()), O, I, KH, KW), "gn_pack_conv_weight")
            ref = want if src.dtype == torch.float32 else packing.pack_conv_weight(src.float())
            assert torch.equal(got, ref), (O, I, KH, KW, src.dtype)
    for (H, K) in ((64, 40), (1280, 320)):
        w, b = torch.randn(2 * H, K, device="cuda") * 0.1, torch.randn(2 * H, device="cuda")
        wp, bp = packing.pack_geglu(w, b)
        gw, gb = torch.empty_like(wp), torch.empty_like(bp)
        check(E.lib.gn_pack_geglu_rows(E._ctx, C.c_void_p(w.data_ptr()), 0, C.c_void_p(gw.data_ptr()), H, K), "gn_pack_geglu_rows")
        check(E.lib.gn_pack_geglu_rows(E._ctx, C.c_void_p(b.data_ptr()), 0, C.c_void_p(gb.data_ptr()), H, 1), "gn_pack_geglu_rows")
        assert torch.equal(gw, wp) and torch.equal(gb, bp)
    N, K = 960, 320
    packed = {"tb.norm1.weight": (1 + 0.2 * torch.randn(K, device="cuda")).half(), "tb.norm1.bias": (0.1 * torch.randn(K, device="cuda")).half(),
              "tb.attn1.to_qkv.weight": (torch.randn(N, K, device="cuda") * K ** -0.5).half()}
    packing.fold_layernorms(packed)
    wg, c1, c2 = torch.empty(N, K, device="cuda", dtype=torch.float16), torch.empty(N, device="cuda"), torch.empty(N, device="cuda", dtype=torch.float16)
    check(E.lib.gn_pack_fold_layernorm(E._ctx, C.c_void_p(packed["tb.attn1.to_qkv.weight"].data_ptr()), C.c_void_p(packed["tb.norm1.weight"].data_ptr()),
                                       C.c_void_p(packed["tb.norm1.bias"].data_ptr()), None, C.c_void_p(wg.data_ptr()), C.c_void_p(c1.data_ptr()),
                                       C.c_void_p(c2.data_ptr()), N, K, K), "gn_pack_fold_layernorm")
    assert torch.equal(wg, packed["tb.attn1.to_qkv.ln_weight"])
    assert float((c1 - packed["tb.attn1.to_qkv.ln_c1"]).abs().max()) <= 1e-5 * float(packed["tb.attn1.to_qkv.ln_c1"].abs().max()) + 1e-6
    assert float((c2.float() - packed["tb.attn1.to_qkv.ln_c2"].float()).abs().max()) <= 2e-3 * float(packed["tb.attn1.to_qkv.ln_c2"].float().abs().max())
