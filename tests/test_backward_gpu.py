"""-m gpu: backward / optimizer kernels of the ControlNet fine-tune step against torch autograd on CPU (fp32, same f16 inputs)."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd import train_ops as T
from genima_amd.packing import pack_conv_weight
from util import assert_close, q16, randn_h, rel_l2

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def h(t):
    return t.half().cuda()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------------------------------- GEMM extensions
def test_gemm_f32_out_accumulate_and_batched(engine):
    M, N, K = 200, 192, 256
    a, w = q16(torch.randn(M, K, generator=g(1))), q16(torch.randn(N, K, generator=g(2)) * K ** -0.5)
    out = torch.full((M, N), 0.5, dtype=torch.float32, device="cuda")
    T.gemm(engine, h(a), h(w), out, M, N, K, K, K, N, f32_out=True, accumulate=True)
    assert_close(out, a @ w.t() + 0.5, what="f32 accumulate gemm")
    # long reduction, few tiles -> the planner picks split-K; the accumulate must still add onto the existing f32 values
    Ml, Kl = 64, 16384
    al, wl = q16(torch.randn(Ml, Kl, generator=g(5))), q16(torch.randn(Ml, Kl, generator=g(6)) * Kl ** -0.5)
    outl = torch.full((Ml, Ml), -2.0, dtype=torch.float32, device="cuda")
    T.gemm(engine, h(al), h(wl), outl, Ml, Ml, Kl, Kl, Kl, Ml, f32_out=True, accumulate=True)
    assert_close(outl, al @ wl.t() - 2.0, what="f32 accumulate gemm (split-K)")
    # batched, two-level strides: [Bo, Bi] problems
    Bo, Bi = 3, 2
    ab = q16(torch.randn(Bo, Bi, M, K, generator=g(3)))
    wb = q16(torch.randn(Bo, Bi, N, K, generator=g(4)) * K ** -0.5)
    ob = torch.empty(Bo, Bi, M, N, dtype=torch.float16, device="cuda")
    T.gemm(engine, h(ab), h(wb), ob, M, N, K, K, K, N, batch=Bo * Bi, batch_inner=Bi, a_bs=(Bi * M * K, M * K), w_bs=(Bi * N * K, N * K),
           out_bs=(Bi * M * N, M * N))
    assert_close(ob, ab @ wb.transpose(-1, -2), what="batched gemm")


def test_linear_backward(engine):
    M, K, N = 1000, 320, 640
    x = q16(torch.randn(M, K, generator=g(1)))
    w = q16(torch.randn(N, K, generator=g(2)) * K ** -0.5)
    dy = q16(torch.randn(M, N, generator=g(3)))
    # dX = dY . W  through the transposed weight copy
    wt = T.transpose2d(engine, h(w), N, K)                       # [K, N]
    dx = torch.empty(M, K, dtype=torch.float16, device="cuda")
    T.gemm(engine, h(dy), wt, dx, M, K, N, N, N, K)
    assert_close(dx, dy @ w, what="linear dX")
    # dW = dY^T . X  (f32, split-K over M) ; db = colsum(dY)
    dyt, xt = T.transpose2d(engine, h(dy), M, N), T.transpose2d(engine, h(x), M, K)
    dw = torch.zeros(N, K, dtype=torch.float32, device="cuda")
    T.gemm(engine, dyt, xt, dw, N, K, (M + 7) // 8 * 8, dyt.stride(0), xt.stride(0), K, f32_out=True, accumulate=True)
    assert_close(dw, dy.t() @ x, what="linear dW")
    db = torch.zeros(N, dtype=torch.float32, device="cuda")
    T.colsum(engine, h(dy), db, 1, M, N, N)
    assert_close(db, dy.sum(0), what="bias grad")
    sh = torch.zeros(4, N, dtype=torch.float32, device="cuda")
    T.colsum(engine, h(dy), sh, 4, M // 4, N, N)
    assert_close(sh, dy.view(4, M // 4, N).sum(1), what="per-batch shift grad")


# ---------------------------------------------------------------------------------------------------- conv backward
def _conv_ref(x, w, stride, pad, up):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    return x, w, F.conv2d(xi, w, None, stride, pad)


@pytest.mark.parametrize("stride,up", [(1, False), (2, False), (1, True)])
def test_conv_backward(engine, stride, up):
    B, Cin, Cout, H = 2, 64, 128, 16
    x0 = q16(torch.randn(B, Cin, H, H, generator=g(1)))
    w0 = q16(torch.randn(Cout, Cin, 3, 3, generator=g(2)) * (9 * Cin) ** -0.5)
    x, w, y = _conv_ref(x0, w0, stride, 1, up)
    dy = q16(torch.randn(y.shape, generator=g(3)))
    y.backward(dy)
    dyn = h(nhwc(dy))
    # dgrad: conv of dY with the 180-degree-rotated, in/out-swapped weights (stride 2: zero-insertion first; upsample: 2x2 sum after)
    wd = pack_conv_weight(w0.flip(2, 3).permute(1, 0, 2, 3).contiguous()).cuda()
    src = T.zero_upsample2x(engine, dyn) if stride == 2 else dyn
    dxi = engine.conv2d(src, wd, None)
    if stride == 2:
        dxi = dxi[:, :H, :H].contiguous()  # the zero-upsampled map is 2*Ho = H here (even H)
    dx = T.sumpool2x2(engine, dxi) if up else dxi
    assert_close(dx, nhwc(x.grad), what=f"conv dgrad s{stride} up{up}")
    # wgrad: dW[co][tap*Cin + ci] = sum_m dY^T[co][m] * im2col^T[tap*Cin + ci][m]
    xin = h(nhwc(F.interpolate(x0, scale_factor=2.0, mode="nearest") if up else x0))
    cols = T.im2col_t(engine, xin, 3, stride, 1)
    M = cols.shape[1]
    dyt = T.transpose2d(engine, dyn.view(M, Cout), M, Cout)
    dw = torch.zeros(Cout, 9 * Cin, dtype=torch.float32, device="cuda")
    T.gemm(engine, dyt, cols, dw, Cout, 9 * Cin, M, M, M, 9 * Cin, f32_out=True, accumulate=True)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    assert_close(dw, ref, what=f"conv wgrad s{stride} up{up}")


# ---------------------------------------------------------------------------------------------------- elementwise / norms
def test_act_geglu_softmax_backward(engine):
    z = q16(torch.randn(64, 128, generator=g(1)) * 2)
    dy = q16(torch.randn(64, 128, generator=g(2)))
    for act, fn in ((1, F.silu), (2, F.gelu), (4, F.relu)):
        zz = z.clone().requires_grad_(True)
        fn(zz).backward(dy)
        assert_close(T.act_bwd(engine, h(dy), h(z), act), zz.grad, what=f"act bwd {act}")
    hg = q16(torch.randn(50, 256, generator=g(3))).requires_grad_(True)
    hid, gate = hg.chunk(2, -1)
    out = hid * F.gelu(gate)
    assert_close(T.geglu_fwd(engine, h(hg.detach())), out.detach(), what="geglu fwd")
    d2 = q16(torch.randn(50, 128, generator=g(4)))
    out.backward(d2)
    assert_close(T.geglu_bwd(engine, h(d2), h(hg.detach())), hg.grad, what="geglu bwd")
    s = (torch.randn(33, 256, generator=g(5)) * 2).requires_grad_(True)
    p = torch.softmax(s * 0.125, -1)
    dp = q16(torch.randn(33, 256, generator=g(6)))
    p16 = q16(p.detach())
    ref = 0.125 * p16 * (dp - (p16 * dp).sum(-1, keepdim=True))
    dpd = h(dp)
    T.softmax_bwd(engine, h(p16), dpd, 0.125)
    assert_close(dpd, ref, what="softmax bwd")


def test_geglu_block_layout_masked_softmax_weight_rotation(engine):
    # packed (32-column interleaved) GEGLU layout == plain layout after un-interleaving
    M, Hd = 40, 128
    hg = q16(torch.randn(M, 2 * Hd, generator=g(1)))
    dy = q16(torch.randn(M, Hd, generator=g(2)))
    inter = torch.stack([hg[:, :Hd].reshape(M, Hd // 32, 32), hg[:, Hd:].reshape(M, Hd // 32, 32)], 2).reshape(M, 2 * Hd)
    assert torch.equal(T.geglu_fwd(engine, h(inter), 32), T.geglu_fwd(engine, h(hg), 0))
    d_plain = T.geglu_bwd(engine, h(dy), h(hg), 0).cpu()
    d_int = T.geglu_bwd(engine, h(dy), h(inter), 32).cpu().reshape(M, Hd // 32, 2, 32)
    assert torch.equal(d_int[:, :, 0].reshape(M, Hd), d_plain[:, :Hd]) and torch.equal(d_int[:, :, 1].reshape(M, Hd), d_plain[:, Hd:])
    # masked softmax: 77 valid keys in an 80-column row
    s = q16(torch.randn(3, 50, 80, generator=g(3)) * 3)
    ref = torch.zeros_like(s)
    ref[..., :77] = torch.softmax(s[..., :77] * 0.125, -1)
    out = T.softmax_rows_masked(engine, h(s), 0.125, 77)
    assert_close(out, ref, what="masked softmax")
    assert float(out[..., 77:].abs().max()) == 0.0
    # data-gradient weights: channel swap + 180-degree tap rotation of a packed 3x3 weight, and the 1x1 case
    w = q16(torch.randn(24, 16, 3, 3, generator=g(4)))
    ref = pack_conv_weight(w.flip(2, 3).permute(1, 0, 2, 3).contiguous())
    assert torch.equal(T.conv_weight_dgrad(engine, pack_conv_weight(w).cuda(), 9).cpu(), ref)
    w1 = q16(torch.randn(24, 16, 1, 1, generator=g(5)))
    assert torch.equal(T.conv_weight_dgrad(engine, pack_conv_weight(w1).cuda(), 1).cpu(), w1[:, :, 0, 0].t().half())
    # padded transpose (GEMM reduction length must be a multiple of 8)
    x = q16(torch.randn(3, 40, generator=g(6)))
    xt = T.transpose2d(engine, h(x), 3, 40)
    assert xt.shape == (40, 8) and torch.equal(xt[:, :3].cpu(), x.t().half()) and float(xt[:, 3:].abs().max()) == 0.0


# (4096, 4096): the self-attention shape of the 64x64-latent level, the one the fine-tune step runs at 512^2
@pytest.mark.parametrize("N,Nk,fused", [(256, 256, True), (200, 77, False), (1024, 1024, True), (64, 77, False), (4096, 4096, True)])
def test_flash_attention_backward(engine, N, Nk, fused):
    """gn_attention_bwd (flash, P recomputed from lse) vs torch autograd of softmax attention; fused = q|k share one buffer."""
    B, heads, D = 2, 3, 64
    Cc = heads * D
    Nkr = (Nk + 7) // 8 * 8
    q = q16(torch.randn(B, N, Cc, generator=g(1)))
    k = torch.zeros(B, Nkr, Cc)
    v = torch.zeros(B, Nkr, Cc)
    k[:, :Nk] = q16(torch.randn(B, Nk, Cc, generator=g(2)))
    v[:, :Nk] = q16(torch.randn(B, Nk, Cc, generator=g(3)))
    dO = q16(torch.randn(B, N, Cc, generator=g(4)))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k[:, :Nk], v[:, :Nk]))
    sp = lambda t, n: t.view(B, n, heads, D).transpose(1, 2)  # noqa: E731
    o_ref = F.scaled_dot_product_attention(sp(qr, N), sp(kr, Nk), sp(vr, Nk)).transpose(1, 2).reshape(B, N, Cc)
    o_ref.backward(dO)
    if fused:
        qk = h(torch.cat([q, k], -1))
        qd, kd, q_off, k_off = qk, qk, 0, Cc
    else:
        qd, kd, q_off, k_off = h(q), h(k), 0, 0
    vd = h(v)
    vt = T.transpose2d(engine, vd, Nkr, Cc, batch=B, in_bs=Nkr * Cc, pad_to=64).view(B, Cc, -1)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    o = engine.attention(qd[:, :, q_off:q_off + Cc], kd[:, :, k_off:k_off + Cc], vt, heads, Nk=Nk, lse=lse)
    assert_close(o, o_ref.detach(), what="attention fwd (lse variant)")
    s = torch.einsum("bnhd,bmhd->bhnm", q.view(B, N, heads, D), k[:, :Nk].view(B, Nk, heads, D)) * D ** -0.5
    assert_close(lse, torch.logsumexp(s, -1) * 1.4426950408889634, what="lse (log2 units)")
    dq = torch.empty_like(qd)
    # NaN-filled: the kernel itself writes the padding rows [Nk, Nkr) of dk / dv as zeros (no fill launch in front of it)
    dk = dq if fused else torch.full_like(kd, float("nan"))
    dv = torch.full_like(vd, float("nan"))
    T.attention_bwd(engine, qd, q_off, kd, k_off, vd, o, h(dO), lse, heads, Nk, dq, dk, dv)
    assert_close(dq[:, :, q_off:q_off + Cc], qr.grad, rel=2e-3, what="dQ")
    assert_close(dk[:, :Nk, k_off:k_off + Cc], kr.grad, rel=2e-3, what="dK")
    assert_close(dv[:, :Nk], vr.grad, rel=2e-3, what="dV")
    if Nkr != Nk:
        assert float(dv[:, Nk:].abs().max()) == 0.0 and float(dk[:, Nk:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,cols,batch,pad_to", [(80, 320, 3, 64), (77, 64, 1, 8), (200, 136, 2, 64), (5, 40, 1, 64), (128, 64, 2, 64)])
def test_transpose_writes_its_own_padding(engine, rows, cols, batch, pad_to):
    """gn_transpose2d_zpad: out[b][c][r] = in[b][r][c] with columns [rows, ld_out) zero -- on a NaN-filled pool so that a column the kernel
    skipped would show (train_ops.transpose2d allocates with torch.empty on this route)."""
    from genima_amd._lib import check

    x = randn_h(batch, rows, cols, seed=rows + cols)
    ld = -(-rows // pad_to) * pad_to
    xt = torch.full((batch, cols, ld), float("nan"), dtype=torch.float16, device="cuda")
    check(engine.lib.gn_transpose2d_zpad(engine._ctx, x.data_ptr(), xt.data_ptr(), rows, cols, cols, ld, batch, rows * cols, cols * ld), "zpad")
    assert torch.equal(xt[:, :, :rows], x.transpose(1, 2))
    assert not bool(torch.isnan(xt).any())
    if ld > rows:
        assert float(xt[:, :, rows:].abs().max()) == 0.0
    via = T.transpose2d(engine, x, rows, cols, batch=batch, in_bs=rows * cols, pad_to=pad_to).view(batch, cols, -1)  # the route the trainer takes
    assert torch.equal(via, xt)


def test_layernorm_backward(engine):
    M, C = 300, 320
    x = q16(torch.randn(M, C, generator=g(1)) * 2 + 0.5).requires_grad_(True)
    gm = q16(1 + 0.1 * torch.randn(C, generator=g(2))).requires_grad_(True)
    bt = q16(0.1 * torch.randn(C, generator=g(3))).requires_grad_(True)
    dy = q16(torch.randn(M, C, generator=g(4)))
    F.layer_norm(x, (C,), gm, bt, 1e-5).backward(dy)
    dgb = torch.zeros(2 * C, dtype=torch.float32, device="cuda")
    dx = T.layernorm_bwd(engine, h(x.detach()), h(gm.detach()), h(dy), dgb[:C], dgb[C:])
    assert_close(dx, x.grad, what="layernorm dx")
    assert_close(dgb[:C], gm.grad, what="layernorm dgamma")
    assert_close(dgb[C:], bt.grad, what="layernorm dbeta")
    # gradient accumulation fused into the kernel: exactly the f16 sum a separate add launch gives
    held = h(q16(torch.randn(M, C, generator=g(5))))
    dx_acc = T.layernorm_bwd(engine, h(x.detach()), h(gm.detach()), h(dy), add=held)
    assert torch.equal(dx_acc, (dx.float() + held.float()).half())


@pytest.mark.parametrize("act,concat", [(1, False), (0, False), (1, True)])
def test_groupnorm_backward(engine, act, concat):
    B, C1, C2, H = 2, 64, (128 if concat else 0), 12
    x1 = q16(torch.randn(B, C1, H, H, generator=g(1)) * 2 + 0.3).requires_grad_(True)
    x2 = q16(torch.randn(B, C2, H, H, generator=g(2))).requires_grad_(True) if concat else None
    C = C1 + C2
    gm = q16(1 + 0.1 * torch.randn(C, generator=g(3))).requires_grad_(True)
    bt = q16(0.1 * torch.randn(C, generator=g(4))).requires_grad_(True)
    xin = torch.cat([x1, x2], 1) if concat else x1
    y = F.group_norm(xin, 32, gm, bt, 1e-5)
    y = F.silu(y) if act else y
    dy = q16(torch.randn(y.shape, generator=g(5)))
    y.backward(dy)
    out, saved = T.groupnorm_fwd_train(engine, h(nhwc(x1.detach())), h(gm.detach()), h(bt.detach()), 32, 1e-5, act,
                                       x2=h(nhwc(x2.detach())) if concat else None)
    assert_close(out, nhwc(y.detach()), what="groupnorm fwd (train)")
    dgb = torch.zeros(2 * C, dtype=torch.float32, device="cuda")
    dx1, dx2 = T.groupnorm_bwd(engine, saved, h(nhwc(dy)), dgamma=dgb[:C], dbeta=dgb[C:])
    assert_close(dx1, nhwc(x1.grad), rel=2e-3, what="groupnorm dx")
    if concat:
        assert_close(dx2, nhwc(x2.grad), rel=2e-3, what="groupnorm dx2")
    assert_close(dgb[:C], gm.grad, rel=2e-3, what="groupnorm dgamma")
    assert_close(dgb[C:], bt.grad, rel=2e-3, what="groupnorm dbeta")
    held1 = h(q16(torch.randn(dx1.shape, generator=g(6))))
    held2 = h(q16(torch.randn(dx2.shape, generator=g(7)))) if concat else None
    a1, a2 = T.groupnorm_bwd(engine, saved, h(nhwc(dy)), add=held1, add2=held2)
    assert torch.equal(a1, (dx1.float() + held1.float()).half())
    if concat:
        assert torch.equal(a2, (dx2.float() + held2.float()).half())


# ---------------------------------------------------------------------------------------------------- loss + optimizer
def test_mse_adamw_clip(engine):
    pred = q16(torch.randn(2, 8, 8, 8, generator=g(1)))
    tgt = q16(torch.randn(2, 8, 8, 4, generator=g(2)))
    p = pred[..., :4].clone().requires_grad_(True)
    loss_ref = F.mse_loss(p, tgt)
    loss_ref.backward()
    loss, dpred = T.mse_loss(engine, h(pred), h(tgt), 4)
    assert abs(float(loss.cpu()) - float(loss_ref)) < 1e-5 * max(1, float(loss_ref))
    assert_close(dpred[..., :4], p.grad, what="mse grad")
    assert float(dpred[..., 4:].abs().max()) == 0.0
    # AdamW, 3 steps, with global-norm clipping computed on the device
    n = 5000
    w = torch.randn(n, generator=g(3)).requires_grad_(True)
    opt = torch.optim.AdamW([w], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    wd = w.detach().clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ss, clip = torch.zeros(1, device="cuda"), torch.zeros(3, device="cuda")
    S = 1024.0  # loss scale carried by the gradients
    for step in range(1, 4):
        gr = torch.randn(n, generator=g(10 + step)) * 3
        w.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_([w], 1.0)
        opt.step()
        gd = (gr * S).cuda()
        T.sumsq(engine, gd, ss)
        T.clip_coef(engine, ss, clip, 1.0, 1.0 / S)
        assert abs(float(clip[1].cpu()) - float(norm)) < 1e-3 * float(norm)
        assert float(clip[2].cpu()) == 0.0
        T.adamw(engine, wd, gd, m, v, 1e-2, 0.9, 0.999, 1e-8, 1e-2, step, clip, 1.0 / S)
    assert rel_l2(wd.cpu(), w.detach()) < 1e-5
    # non-finite gradients: flagged, and the step leaves parameters and moments untouched
    before = (wd.clone(), m.clone(), v.clone())
    gd[7] = float("inf")
    T.sumsq(engine, gd, ss)
    T.clip_coef(engine, ss, clip, 1.0, 1.0 / S)
    T.adamw(engine, wd, gd, m, v, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 4, clip, 1.0 / S)
    assert float(clip[2].cpu()) == 1.0
    assert torch.equal(wd, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
    out16 = torch.empty(n, dtype=torch.float16, device="cuda")
    T.cast_f32_f16(engine, wd, out16)
    assert torch.equal(out16.cpu(), wd.cpu().half())


@pytest.mark.parametrize("rows,cols,groups", [(4096, 320, 8), (1024, 136, 1), (640, 64, 5), (1000, 72, 1)])
def test_transpose_with_column_sums(engine, rows, cols, groups):
    """gn_transpose2d_colsum: x^T and the bias (1 group) / per-sample time-shift (batch groups) gradients from one pass over dY
    (autograd's `.sum(0)` of the Linear / conv bias backward); (1000, 72): a row block that is not 64-row tiled takes the two-launch path."""
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(rows, cols, generator=g) * 0.7).half().cuda()
    s1 = torch.full((1, cols), 0.5, device="cuda")
    sg = torch.full((groups, cols), -0.25, device="cuda")
    xt = T.transpose2d_colsum(engine, x, rows, cols, [(sg, groups), (None, 3), (s1, 1)])
    xf = x.float().cpu()
    assert torch.equal(xt[:, :rows].cpu(), x.t().cpu())
    assert_close(sg, xf.view(groups, rows // groups, cols).sum(1) - 0.25, rel=1e-5, what="grouped column sums")
    assert_close(s1, xf.sum(0, keepdim=True) + 0.5, rel=1e-5, what="column sums")
    xt2 = T.transpose2d_colsum(engine, x, rows, cols, [])
    assert torch.equal(xt2, xt)


@pytest.mark.parametrize("R,N,K,tile", [(4096, 320, 320, 0), (1000, 136, 72, 0), (8192, 1280, 320, 1), (777 * 8, 64, 640, 2), (32768, 320, 1280, 0),
                                        (8192, 1280, 320, 3), (32768, 320, 1280, 3), (777 * 8, 136, 264, 3), (8192, 1280, 320, 4), (4096, 320, 328, 4)])  # 128 x 256 / 256 x 128 on eight waves
def test_wgrad_linear_natural_layout(engine, R, N, K, tile):
    """gn_wgrad (csrc/gemm_tn.hip): dW += dY^T X from row-major dY [R, N] and X [R, K] -- LDS transpose reads instead of transposed copies
    -- against an fp64 torch product of the same f16 inputs; accumulation into existing f32 values; ragged R / N / K tails."""
    gg = torch.Generator().manual_seed(R + N)
    dy = (torch.randn(R, N, generator=gg) * 0.5).half().cuda()
    x = (torch.randn(R, K, generator=gg) * 0.5).half().cuda()
    dw = torch.full((N, K), 0.125, device="cuda")
    T.wgrad(engine, dy, x, dw, tile=tile)
    ref = (dy.double().t() @ x.double()).float().cpu() + 0.125
    assert_close(dw, ref, rel=2e-4, what=f"wgrad {R}x{N}x{K}")
    first = dw.clone()
    T.wgrad(engine, dy, x, dw, tile=tile)
    assert_close(dw, 2 * ref - 0.125, rel=2e-4, what="wgrad accumulate")
    dw2 = torch.full((N, K), 0.125, device="cuda")
    db = torch.full((N,), 2.0, device="cuda")
    groups = 8 if R % (8 * 64) == 0 else 0
    ds = torch.full((groups, N), -1.0, device="cuda") if groups else None
    T.wgrad(engine, dy, x, dw2, tile=tile, dbias=db, dshift=ds, shift_groups=groups)
    if not groups:
        assert torch.equal(dw2, first), "deterministic"  # (per-sample sums re-slice the rows: same values, another summation order)
    assert_close(dw2, ref, rel=2e-4, what="wgrad with column sums")
    assert_close(db, dy.double().sum(0).float().cpu() + 2.0, rel=1e-4, what="bias gradient from the dY fragments")
    if groups:
        assert_close(ds, dy.double().view(groups, R // groups, N).sum(1).float().cpu() - 1.0, rel=1e-4, what="per-sample shift gradient")


@pytest.mark.parametrize("B,H,C,N,ks,stride", [(2, 16, 64, 72, 3, 1), (2, 32, 128, 128, 3, 1), (3, 16, 320, 320, 3, 2), (2, 12, 64, 64, 1, 1),
                                               (8, 32, 640, 640, 3, 1), (2, 64, 16, 32, 3, 2), (2, 32, 96, 256, 3, 2), (1, 48, 8, 16, 3, 1),
                                               # feature maps smaller than one 64-row K tile: 4 x 4 and 2 x 2 outputs, several tiles per slice
                                               (8, 4, 64, 64, 3, 1), (8, 4, 128, 64, 1, 1), (12, 8, 64, 96, 3, 2), (40, 2, 32, 32, 3, 1), (9, 6, 64, 64, 3, 1)])
def test_wgrad_conv_natural_layout(engine, B, H, C, N, ks, stride):
    """Conv weight gradient straight from NHWC x and dY (no im2col^T): against autograd's conv2d weight gradient (fp32 on the f16 inputs),
    in the packed [Cout, tap * C + c] layout of the forward weights."""
    gg = torch.Generator().manual_seed(B * H + C)
    x = q16(torch.randn(B, C, H, H, generator=gg) * 0.5)
    w = torch.zeros(N, C, ks, ks, requires_grad=True)
    y = F.conv2d(x, w, None, stride=stride, padding=ks // 2)
    dy = q16(torch.randn(y.shape, generator=gg) * 0.5)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, ks * ks * C)  # [Cout, (dy, dx, c)]
    dw = torch.zeros(N, ks * ks * C, device="cuda")
    T.wgrad(engine, h(nhwc(dy)), h(nhwc(x)), dw, ksize=ks, stride=stride, pad=ks // 2)
    assert_close(dw, ref, rel=5e-4, what=f"conv wgrad {C}->{N} k{ks} s{stride}")
    for tile in (3, 4):  # the eight-wave tiles (128 x 256, 256 x 128) where the gradient is at least one tile wide / tall
        if (tile == 3 and N >= 128 and ks * ks * C >= 256) or (tile == 4 and N >= 256 and ks * ks * C >= 128):
            dw = torch.zeros(N, ks * ks * C, device="cuda")
            T.wgrad(engine, h(nhwc(dy)), h(nhwc(x)), dw, ksize=ks, stride=stride, pad=ks // 2, tile=tile)
            assert_close(dw, ref, rel=5e-4, what=f"conv wgrad {C}->{N} k{ks} s{stride} tile {tile}")


def test_transpose2d_multi_matches_single_launches(engine):
    """gn_transpose2d_multi: a table of Linear weights (W^T) and packed 3x3 conv weights (tap-rotated data-gradient form) in one launch,
    bit for bit what the separate gn_transpose2d launches write."""
    import ctypes as C
    from genima_amd import train_ops as T
    from genima_amd._lib import check
    ws = [(randn_h(320, 1280, seed=1), 0), (randn_h(640, 9 * 320, seed=2), 9), (randn_h(72, 200, seed=3), 0), (randn_h(128, 4 * 64, seed=4), 4)]
    refs, outs, rows, blocks = [], [], [], 0
    for w, taps in ws:
        if taps:
            Cout, K = w.shape
            Cin = K // taps
            refs.append(T.conv_weight_dgrad(engine, w, taps))
            out = torch.zeros_like(refs[-1])
            it = (w.data_ptr(), out.data_ptr() + 2 * (taps - 1) * Cout, taps * Cin, taps * Cout, Cin, -Cout, Cout, Cin, taps)
        else:
            N, K = w.shape
            refs.append(T.transpose2d(engine, w, N, K))
            out = torch.zeros_like(refs[-1])
            it = (w.data_ptr(), out.data_ptr(), K, out.stride(0), 0, 0, N, K, 1)
        src, dst, ld_in, ld_out, in_bs, out_bs, r, c, batch = it
        rows.append([src, dst, ld_in, ld_out, in_bs, out_bs, (r & 0xFFFFFFFF) | (c << 32), (batch & 0xFFFFFFFF) | (blocks << 32)])
        blocks += batch * (-(-r // 64)) * (-(-c // 64))
        outs.append(out)
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    check(engine.lib.gn_transpose2d_multi(engine._ctx, table.data_ptr(), len(rows), blocks), "gn_transpose2d_multi")
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
